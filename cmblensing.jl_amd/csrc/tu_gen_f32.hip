// the any-size transform launches in float precision: k_ct_*, k_gen_dft* (engine_gen.hpp; api_decl.hpp has the map of the build)
#include "engine_gen.hpp"
namespace cmbl { CMBL_INSTANTIATE_GEN(float) }
CMBL_STAMPS_READER(gen_f32)
