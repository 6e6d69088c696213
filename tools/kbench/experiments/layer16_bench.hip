// Standalone timing harness for the recurrent-layer kernel (development tool, not part of the product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DTD_ABLATE_*] [
//   ./lb [batch] [L] [reps] [ref.bin to write | to compare]
#ifndef KERNEL_HDR
#define KERNEL_HDR "rnn_layer16.h"
#endif
#include KERNEL_HDR
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include <algorithm>

#ifndef NU_
#define NU_ 8
#endif
#ifndef KX_
#define KX_ 16
#endif
#ifndef L0_
#define L0_ false
#endif
#ifndef GRU_
#define GRU_ false
#endif
#ifndef HEAD_
#define HEAD_ 0
#endif

void td_set_error(const char *, ...) {}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static uint16_t bf16_bits(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf16_val(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 16384, L = argc > 2 ? atoi(argv[2]) : 272, reps = argc > 3 ? atoi(argv[3]) : 3;
    const char *mode = argc > 4 ? argv[4] : "";   // "write" or "check"
    const char *path = argc > 5 ? argv[5] : "/tmp/layer_ref.bin";
    using T = Td16Traits<NU_, KX_, L0_, GRU_, HEAD_>;
    const int nbb = (B + 31) / 32, dirs = 2;
    const size_t dir_bytes = (size_t)((T::NF + TD16_RING_FRAGS - 1) / TD16_RING_FRAGS) * TD16_RING_BYTES;
    const int in_frags = T::XFR, out_frags = dirs * 2 * NU_;
    const size_t wbytes = dir_bytes * dirs, xbytes = (size_t)L * nbb * in_frags * TD_FRAG_BYTES, obytes = (size_t)L * nbb * out_frags * TD_FRAG_BYTES;
    std::vector<uint16_t> hw(wbytes / 2), hx(xbytes / 2);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f) * 2.f - 1.f; };
    for (auto &v : hw) v = bf16_bits(rnd() * (1.0f / 16.0f));
    for (auto &v : hx) v = bf16_bits(rnd());
    char *dw, *dx, *dout;
    CK(hipMalloc(&dw, wbytes)); CK(hipMalloc(&dx, xbytes)); CK(hipMalloc(&dout, obytes));
    CK(hipMemcpy(dw, hw.data(), wbytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, hx.data(), xbytes, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0, obytes));
    TdLayerParams p;
    memset(&p, 0, sizeof(p));
    p.wpacked = dw; p.dir_stride = dir_bytes; p.xin = dx; p.out = dout; p.in_frags = in_frags; p.out_frags = out_frags;
    p.L = L; p.nbb = nbb; p.dirs = dirs; p.save = nullptr; p.save_dir_stride = 0;
    float *dhp = nullptr;
    if (HEAD_) { CK(hipMalloc(&dhp, (size_t)L * nbb * dirs * 64 * 20 * sizeof(float))); p.hpart = dhp; p.hp_stride = HEAD_ == 1 ? 12 : 20; }
    const int tiles = (nbb + TD_BLOCK_WAVES - 1) / TD_BLOCK_WAVES;
#ifdef TD_STAMP
    unsigned long long *dst;
    CK(hipMalloc(&dst, (size_t)tiles * dirs * 16));
    p.save = (float *)dst;
#endif
    auto launch = [&]() { hipLaunchKernelGGL((td_rnn_layer16_kernel<NU_, KX_, L0_, GRU_, HEAD_>), dim3(tiles * dirs), dim3(256), 0, 0, p); };
    launch();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; sum += ms;
    }
    const double mfma_per_wave_step = (double)T::NF;  /* fragments: two 16x16x32 each = one 32x32x16 of work */
    const double flops = (double)nbb * dirs * L * mfma_per_wave_step * 32768.0;
    printf("%s B=%d L=%d NU=%d KX=%d L0=%d GRU=%d: best %.3f ms avg %.3f ms  %.1f TFLOP/s(issued)  %.1f cyc/MFMA@2.4GHz\n", argv[0], B, L, NU_, KX_,
           (int)L0_, (int)GRU_, best, sum / reps, flops / (best * 1e-3) / 1e12,
           best * 1e-3 * 2.4e9 / (L * mfma_per_wave_step * ((tiles * dirs + 255) / 256)));
#ifdef TD_STAMP
    {
        std::vector<unsigned long long> st((size_t)tiles * dirs * 2);
        CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> ghz;
        double cyc = 0;
        for (size_t i = 0; i < st.size(); i += 2) { ghz.push_back((double)st[i] / (double)st[i + 1] * 0.1); cyc += (double)st[i]; }
        std::sort(ghz.begin(), ghz.end());
        printf("   in-kernel clock: median %.3f GHz (min %.3f max %.3f), mean cycles/WG %.0f -> %.1f cyc/MFMA(actual)\n", ghz[ghz.size() / 2], ghz.front(),
               ghz.back(), cyc / (st.size() / 2), cyc / (st.size() / 2) / (L * mfma_per_wave_step));
    }
#endif
    if (mode[0]) {
        std::vector<uint16_t> ho(obytes / 2);
        CK(hipMemcpy(ho.data(), dout, obytes, hipMemcpyDeviceToHost));
        if (!strcmp(mode, "write")) {
            FILE *f = fopen(path, "wb"); fwrite(ho.data(), 1, obytes, f); fclose(f);
            printf("wrote %s\n", path);
        } else {
            std::vector<uint16_t> ref(obytes / 2);
            FILE *f = fopen(path, "rb");
            if (!f || fread(ref.data(), 1, obytes, f) != obytes) { printf("cannot read %s\n", path); return 1; }
            fclose(f);
            double maxd = 0; size_t nbad = 0, nnz = 0;
            for (size_t i = 0; i < ho.size(); ++i) {
                float a = bf16_val(ho[i]), b = bf16_val(ref[i]);
                double d = fabs((double)a - b);
                if (d > maxd) maxd = d;
                if (d > 0.02) ++nbad;
                if (ho[i]) ++nnz;
            }
            printf("check vs %s: max |diff| = %.5f, > 0.02: %zu of %zu, nonzero %zu  %s\n", path, maxd, nbad, ho.size(), nnz, nbad == 0 && nnz > ho.size() / 2 ? "OK" : "MISMATCH");
        }
    }
    return 0;
}
