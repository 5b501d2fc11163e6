// every typed body behind the C ABI in float precision, and with them Ctx / Flow / Dataset / Drivers and their kernels (api_decl.hpp)
#include "api_body.hpp"
namespace cmbl { CMBL_INSTANTIATE_API(float) }
CMBL_STAMPS_READER(main_f32)
