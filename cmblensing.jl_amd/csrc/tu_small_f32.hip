// the small-map flows in float precision: k_small_flow, k_small_adj (engine_small.hpp; api_decl.hpp has the map of the build)
#include "engine_small.hpp"
namespace cmbl { CMBL_INSTANTIATE_SMALL(float) }
CMBL_STAMPS_READER(small_f32)
