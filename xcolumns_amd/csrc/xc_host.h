// xc_host.h -- host-side helpers shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "xcolumns_amd.h"

namespace xc {

// thread-local text of the last failure, returned by xc_last_error()
char *err_buf();
int fail_arg(int code, const char *fmt, ...);
int fail_hip(hipError_t e, const char *what);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// chunks of 64 candidates a lane must hold for rows of up to max_row_nnz
// entries; 0 when the row is too long for the register-resident kernels
inline int chunks_for(int max_row_nnz) {
    if (max_row_nnz <= 64) return 1;
    if (max_row_nnz <= 128) return 2;
    if (max_row_nnz <= 256) return 4;
    if (max_row_nnz <= 512) return 8;
    if (max_row_nnz <= 1024) return 16;
    return 0;
}

// number of wavefronts to launch for an embarrassingly parallel row loop
int default_row_waves(int64_t n_rows);

} // namespace xc

#define XC_HIP_TRY(expr)                                         \
    do {                                                         \
        hipError_t _e = (expr);                                  \
        if (_e != hipSuccess) return xc::fail_hip(_e, #expr);    \
    } while (0)

#define XC_CHECK_LAUNCH(name)                                    \
    do {                                                         \
        hipError_t _e = hipGetLastError();                       \
        if (_e != hipSuccess) return xc::fail_hip(_e, name);     \
    } while (0)
