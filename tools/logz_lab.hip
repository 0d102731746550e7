// logz_lab.hip -- per-kernel timing of the logZ pipeline with HIP events, no Python
// (tools only).  Rebuild with -D knobs to try variants:
//
//   hipcc --offload-arch=gfx950 -O3 -Itaiyaki_amd/csrc [-DTK_K3_NT_STORE=0 ...] \
//         tools/logz_lab.hip -o tools/logz_lab && tools/logz_lab [T N [CH]]
#include "../taiyaki_amd/csrc/logz_kernels.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

using namespace tk;

template <typename F>
static double timeit(F launch, int reps = 21) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ms;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[reps / 2] * 1e3;
}

template <int CH>
static void run(const float *scores, int T, int N, float *logz, float *grad, void *wsmem, uint32_t *status) {
    constexpr int NB = 4;
    using F = FF<NB>;
    LogzWs ws;
    logz_ws_layout<NB>(T, N, wsmem, &ws);
    ws.nstride = N;             // (row stride of the tensor; left at 0 every row is row 0 and the op looks 20 % faster than it is)
    ws.grad_scale = 1.f;
    ws.grad_scale_vec = nullptr;
    const int C = (T + CH - 1) / CH, SUP = logz_super(C), NSUP = (C + SUP - 1) / SUP;
    const int ncols = (N + WAVE - 1) / WAVE, Npad = ncols * WAVE;
    const size_t lds1 = K1_WAVES * std::max(4 * (size_t)XMat<NB>::NF4 * WAVE, 4 * (size_t)WAVE * F::PIECES) * sizeof(float);
    size_t lds2 = logz_middle_lds_bytes<NB>(C, NSUP);
    if (getenv("LAB_LDS2")) lds2 = std::max(lds2, (size_t)atoi(getenv("LAB_LDS2")));
    constexpr bool chain_in_buf = ((CH / K3_WAVES) + 2) * F::NS * WAVE <= k3_buf_f4<NB, CH>() * 4;
    const size_t lds3 = K3_WAVES * (size_t)k3_buf_f4<NB, CH>() * sizeof(f4) + (chain_in_buf ? 0 : 2 * F::NS * WAVE * sizeof(float));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&logz_transfer_kernel<NB, CH, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&logz_transfer_kernel<NB, CH, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&logz_transfer_kernel<NB, CH, 3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const bool ring = getenv("LAB_RING") ? atoi(getenv("LAB_RING")) != 0 : (size_t)ncols * C <= 900;
    const size_t ringlds = std::max(lds1, K1_WAVES * (size_t)3 * WAVE * F::PIECES * sizeof(f4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&logz_middle_kernel<NB, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&logz_middle_kernel<NB, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&logz_posterior_kernel<NB, CH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&logz_transfer_coop_kernel<NB, CH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int nt1 = getenv("LAB_NT1") ? atoi(getenv("LAB_NT1")) : ((size_t)T * N * 160 > ((size_t)300 << 20));
    const int nt3 = getenv("LAB_NT3") ? atoi(getenv("LAB_NT3")) : ((size_t)T * N * 160 > ((size_t)200 << 20));
    auto k1 = [&] {
        if ((size_t)ncols * C >= 640 && ring) hipLaunchKernelGGL((logz_transfer_kernel<NB, CH, 3, false>), dim3(ncols, (C + K1_WAVES - 1) / K1_WAVES), dim3(K1_WAVES * WAVE), ringlds, 0, scores, T, N, C, Npad, ws);
        else if ((size_t)ncols * C >= 640 && nt1) hipLaunchKernelGGL((logz_transfer_kernel<NB, CH, 0, true>), dim3(ncols, (C + K1_WAVES - 1) / K1_WAVES), dim3(K1_WAVES * WAVE), lds1, 0, scores, T, N, C, Npad, ws);
        else if ((size_t)ncols * C >= 640) hipLaunchKernelGGL((logz_transfer_kernel<NB, CH, 0, false>), dim3(ncols, (C + K1_WAVES - 1) / K1_WAVES), dim3(K1_WAVES * WAVE), lds1, 0, scores, T, N, C, Npad, ws);
        else hipLaunchKernelGGL((logz_transfer_coop_kernel<NB, CH>), dim3(ncols, C), dim3(K1_WAVES * WAVE), lds1, 0, scores, T, N, C, Npad, ws);
    };
    auto k2 = [&] {
        if (SUP == 8) hipLaunchKernelGGL((logz_middle_kernel<NB, 8>), dim3(N), dim3(K2_WAVES * WAVE), lds2, 0, N, C, NSUP, Npad, ws, logz, 1, status);
        else hipLaunchKernelGGL((logz_middle_kernel<NB, 16>), dim3(N), dim3(K2_WAVES * WAVE), lds2, 0, N, C, NSUP, Npad, ws, logz, 1, status);
    };
    auto k3 = [&] { hipLaunchKernelGGL((logz_posterior_kernel<NB, CH>), dim3(ncols, C), dim3(K3_WAVES * WAVE), lds3, 0, scores, grad, T, N, Npad, ws, status, nt3); };
    k1(); k2(); k3();
    CK(hipDeviceSynchronize());
    const double a = timeit(k1), b = timeit(k2), c = timeit(k3), all = timeit([&] { k1(); k2(); k3(); });
    if (getenv("LAB_SPREAD")) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        printf("   per-launch us:");
        for (int i = 0; i < 40; ++i) {
            CK(hipEventRecord(e0, 0)); k1(); k2(); k3(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); printf(" %.1f", t * 1e3);
        }
        printf("\n");
    }
    const double alg = 3.0 * T * N * F::S * 4;
    float z0;
    CK(hipMemcpy(&z0, logz, 4, hipMemcpyDeviceToHost));
    printf("T=%d N=%d CH=%d  transfer %6.1f  middle %6.1f  posterior %6.1f  sum %6.1f | back-to-back %6.1f us -> %5.2f TB/s (%4.1f%% of 8)  logz[0]=%.4f\n",
           T, N, CH, a, b, c, a + b + c, all, alg / all / 1e6, alg / all / 1e6 / 8 * 100, z0);
#ifdef TK_LAB_TIMING
    k2();
    CK(hipDeviceSynchronize());
    long long st[8];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(tk_dbg), sizeof st));
    k1();
    CK(hipDeviceSynchronize());
    long long s1[16];
    CK(hipMemcpyFromSymbol(s1, HIP_SYMBOL(tk_dbg), sizeof s1));
    printf("   transfer (block (1,20) wave 0, shader clocks): first row done %lld  rows 1-8 %lld  rows 9-16 %lld  store %lld\n", s1[9] - s1[8], s1[10] - s1[9], s1[11] - s1[10], s1[12] - s1[11]);
    printf("   middle phases (block 100, shader clocks): stage %lld  combine %lld  scan %lld  expand %lld\n", st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3]);
#endif
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 4000, N = argc > 2 ? atoi(argv[2]) : 256;
    const int ch = argc > 3 ? atoi(argv[3]) : 0;
    const size_t n = (size_t)T * N * 40;
    std::vector<float> h(n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h[i] = (float)((s >> 40) * (1.0 / 16777216.0) * 10.0 - 5.0);
    }
    float *scores, *grad, *logz; void *ws; uint32_t *status;
    CK(hipMalloc(&scores, n * 4)); CK(hipMalloc(&grad, n * 4)); CK(hipMalloc(&logz, N * 4)); CK(hipMalloc(&status, 4));
    const size_t wsb = logz_workspace_bytes(T, N, 4);
    CK(hipMalloc(&ws, wsb));
    CK(hipMemcpy(scores, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(status, 0, 4));
    if (ch == 0 || ch == 32) run<32>(scores, T, N, logz, grad, ws, status);
    if (ch == 0 || ch == 16) run<16>(scores, T, N, logz, grad, ws, status);
    if (ch == 0 || ch == 8) run<8>(scores, T, N, logz, grad, ws, status);
    return 0;
}
