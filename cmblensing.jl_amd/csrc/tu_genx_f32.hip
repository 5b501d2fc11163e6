// the x-side any-size launches in float precision: k_ct_dft2, k_ct_adj_x, k_ct_adj_x_dx (engine_gen.hpp; api_decl.hpp has the map of the build)
#include "engine_gen.hpp"
namespace cmbl { CMBL_INSTANTIATE_GENX(float) }
CMBL_STAMPS_READER(genx_f32)
