// rt_xcheck_host_ctx.hpp -- part of the CROSS-CHECK build only (-DRTIOW_CROSSCHECK_MODES: tools/librtiow_hip_xcheck.so, a test artefact).
// Host side of scan modes 2-4, the earlier matrix-pipe forms of the sphere-scan filter; the product library carries
// modes 0, 1 and 5 and never includes this file.  The device tables of modes 2-4, a member of rt_context.
#pragma once
#include <hip/hip_runtime.h>
struct XcheckScene {
    float *d_bmat = nullptr;       // [tiles][64] MFMA B operand of the filter
    float *d_kpt = nullptr;        // [tiles][16] K' per sphere
    uint4 *d_bmat16 = nullptr;     // [tiles][64] bf16x3 B operand
    float *d_kpt16 = nullptr;      // [tiles][16] K' for the bf16x3 form
    uint4 *d_bmatL = nullptr;      // [tiles][2][64] MODE 4 (lifted form) B operands
    void release()
    {
        (void)hipFree(d_bmat); (void)hipFree(d_kpt); (void)hipFree(d_bmat16); (void)hipFree(d_kpt16); (void)hipFree(d_bmatL);
        d_bmat = d_kpt = nullptr; d_bmat16 = nullptr; d_kpt16 = nullptr; d_bmatL = nullptr;
    }
};
