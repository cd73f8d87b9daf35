// ORACLE — test infrastructure only (see rotated_iou.c).  The hull ordering of detectron2's CPU path, box_iou_rotated_utils.h
// convex_hull_graham, "#else // CPU version": std::sort over the translated points with a polar-angle comparator that falls back to the
// distance for |cross| < 1e-6.  That comparator is not a strict weak order when several points are nearly collinear with the pivot
// (near-duplicate boxes), so what std::sort returns is implementation-defined; this file IS this toolchain's std::sort (libstdc++:
// insertion sort up to 16 elements, introsort above), which is what a detectron2 CPU build on this image would run.
// Built only to compare the two orderings on the fuzz families (tests/test_iou_fuzz.py); the parity contract is the CUDA path's
// exchange sort in rotated_iou.c.
#include <algorithm>
#include <cmath>

struct pt_t { float x, y; };

extern "C" void ora_hull_stdsort(pt_t* q, int n)
{
    std::sort(q, q + n, [](const pt_t& A, const pt_t& B) -> bool {
        const float temp = A.x * B.y - B.x * A.y;
        if (std::fabs((double)temp) < 1e-6) return (A.x * A.x + A.y * A.y) < (B.x * B.x + B.y * B.y);
        return temp > 0;
    });
}
