while read h ci co n; do
  for ms in 24 48; do
  b=$(CHECK=0 W3=1 RYOLO_W3_FORCE=1 RYOLO_W3_MINSTEPS=$ms timeout 120 python tools/bench_wgrad.py 64 $h $ci $co 3 1 10 2>&1 | grep -E "wgrad" | sed 's/.*splitk/splitk/')
  echo "H$h $ci->$co x$n | minsteps $ms W3 $b"
  done
done <<LIST
100 128 64 1
100 64 64 3
50 256 128 2
50 128 128 6
25 512 256 1
25 256 256 7
25 512 512 2
LIST
