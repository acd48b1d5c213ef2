// The T3D_X3 instantiations of the per-point GEMM kernels (fp32 layers on the bf16 matrix pipe, three bf16 terms per operand):
// csrc/pointmlp.hip compiled with T3D_X3_TU, i.e. its device templates and the t3d_x3_* launch helpers without the entry points.
#define T3D_X3_TU 1
#include "pointmlp.hip"
