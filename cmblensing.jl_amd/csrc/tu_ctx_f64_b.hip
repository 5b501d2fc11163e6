// compile-time-plan kernels, double precision, lengths of CMBL_CT_LIST_B: row side of the fused stages (k_ct_adj_x, k_ct_adj_x_dx, k_ct_dft2) (engine_ct.hpp)
#include "engine_ct.hpp"
namespace cmbl {
#define CMBL_X(n) template struct CtLaunchX<double, n>;
CMBL_CT_LIST_B(CMBL_X)
#undef CMBL_X
}
CMBL_STAMPS_READER(ctx_f64_b)
