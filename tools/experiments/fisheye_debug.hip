// Debug harness: one failing ComputeStereoFishEyeMatches case through the device functions of orbx_match.hip, intermediates printed as
// hex floats (compare with the numpy evaluation of the same steps).  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off
#include "../../wut-cuda-orb-slam3_amd/csrc/orbx_match.hip"
#include <cstdio>
#include <cstring>
#include "fisheye_case.inc"
__global__ void k_dbg(FisheyeArgs a, float x1, float y1, float x2, float y2, float *out)
{
    float r1[3], r2[3];
    fe_unproject(a.cam1, a.precision, x1, y1, r1);
    fe_unproject(a.cam2, a.precision, x2, y2, r2);
    out[0] = r1[0]; out[1] = r1[1]; out[2] = r2[0]; out[3] = r2[1];
    float p[3];
    out[4] = fe_triangulate(a, x1, y1, x2, y2, 1.f, 1.f, p);
    out[5] = p[0]; out[6] = p[1]; out[7] = p[2];
    // the pieces of unproject
    const float *P = a.cam1;
    const float pwx = __fdiv_rn(__fsub_rn(x1, P[2]), P[0]), pwy = __fdiv_rn(__fsub_rn(y1, P[3]), P[1]);
    out[8] = pwx; out[9] = pwy; out[10] = __fsqrt_rn(__fadd_rn(__fmul_rn(pwx, pwx), __fmul_rn(pwy, pwy)));
    out[11] = (float)tan((double)1.2879611f);
}
int main()
{
    FisheyeArgs a;
    memset(&a, 0, sizeof a);
    for (int i = 0; i < 8; ++i) { a.cam1[i] = C1[i]; a.cam2[i] = C2[i]; }
    for (int i = 0; i < 9; ++i) a.R12[i] = R[i];
    for (int i = 0; i < 3; ++i) a.t12[i] = T[i];
    a.precision = 1e-6f;
    float *d; hipMalloc(&d, 64 * 4);
    hipLaunchKernelGGL(k_dbg, dim3(1), dim3(1), 0, 0, a, X1, Y1, X2, Y2, d);
    float h[16]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 12; ++i) printf("out[%d] = %a (%.9g)\n", i, h[i], h[i]);
    return 0;
}
