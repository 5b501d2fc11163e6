// the host side of the any-size transform launches in double precision and the run-time-plan kernels k_gen_dft* (engine_gen.hpp; the
// compile-time-plan kernels are compiled by tu_cty_* / tu_ctx_*; api_decl.hpp has the map of the build)
#include "engine_gen.hpp"
namespace cmbl { CMBL_INSTANTIATE_GEN(double) CMBL_INSTANTIATE_GENX(double) }
CMBL_STAMPS_READER(gen_f64)
