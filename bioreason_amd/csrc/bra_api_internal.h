// bra_api_internal.h — status codes shared by the C-ABI entry points.
#pragma once
// 0 = ok; > 0 = hipError_t from the launch; < 0 = argument error (never throws).
#define BRA_OK 0
#define BRA_ERR_ARG (-1)
#define BRA_ERR_UNSUPPORTED (-2)
