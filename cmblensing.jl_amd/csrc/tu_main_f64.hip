// every typed body behind the C ABI in double precision, and with them Ctx / Flow / Dataset / Drivers and their kernels (api_decl.hpp)
#include "api_body.hpp"
namespace cmbl { CMBL_INSTANTIATE_API(double) }
CMBL_STAMPS_READER(main_f64)
