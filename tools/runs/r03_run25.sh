bash tools/collect_evidence.sh > gpurun_out/r03m_log.txt 2>&1
tail -n 30 gpurun_out/r03m_log.txt
