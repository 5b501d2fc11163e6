// compile-time-plan kernels, double precision, lengths of CMBL_CT_LIST_B: column side of the fused stages + the plain transforms (k_ct_dft, k_ct_dftx, k_ct_flow_y, k_ct_delta_y, k_ct_adj_y) (engine_ct.hpp)
#include "engine_ct.hpp"
namespace cmbl {
#define CMBL_X(n) template struct CtLaunchY<double, n>;
CMBL_CT_LIST_B(CMBL_X)
#undef CMBL_X
}
CMBL_STAMPS_READER(cty_f64_b)
