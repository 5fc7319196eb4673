// rt_xcheck_params.hpp -- part of the CROSS-CHECK build only (-DRTIOW_CROSSCHECK_MODES: tools/librtiow_hip_xcheck.so, a test artefact).
// Scan modes 2-4 are the earlier matrix-pipe forms of the sphere-scan filter; the product library carries modes 0, 1
// and 5 and never includes this file.  The device tables of modes 2-4, a member of rt::KParams.
#pragma once
#include <hip/hip_runtime.h>
namespace rt {
struct KXcheckTables {
    const float *bmat;         // [tiles][64] MFMA B operand: lane l -> S[k=l>>4][sphere 16t+(l&15)], S=(cx,cy,cz,1)
    const float *kpt;          // [tiles][16] K' per sphere (NaN: never kept by the filter)
    const uint4 *bmat16;       // [tiles][64] bf16x3 B operand: 8 bf16 (y1,y2,y1,y3,y2,y1,y3,y2) of S[k][sphere]
    const float *kpt16;        // [tiles][16] K' for the bf16x3 form (larger KU)
    const uint4 *bmatL;        // [tiles][2][64] MODE 4 B operands: the 64 K-slots of the lifted form (rt_device.hpp)
};
} // namespace rt
