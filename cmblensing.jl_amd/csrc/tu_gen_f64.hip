// the any-size transform launches in double precision: k_ct_*, k_gen_dft* (engine_gen.hpp; api_decl.hpp has the map of the build)
#include "engine_gen.hpp"
namespace cmbl { CMBL_INSTANTIATE_GEN(double) }
CMBL_STAMPS_READER(gen_f64)
