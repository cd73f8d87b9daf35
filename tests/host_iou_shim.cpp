// Host build of the PRODUCT's pair function (r-yolov4_amd/csrc/rotated_iou.h) so its restructured arithmetic
// (per-box prologue + pair function) can be checked bit-for-bit against oracle/rotated_iou.c without a GPU.
#include "../r-yolov4_amd/csrc/rotated_iou.h"
extern "C" float host_pair_iou(const float* b1, const float* b2)
{
    BoxPrep A, B;
    box_prep(b1, A);
    box_prep(b2, B);
    return rotated_iou_pair(A, B);
}
extern "C" int host_far_apart(const float* b1, const float* b2)
{
    BoxPrep A, B;
    box_prep(b1, A);
    box_prep(b2, B);
    return boxes_far_apart(A, B) ? 1 : 0;
}
