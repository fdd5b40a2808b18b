// Stand-alone timing harness of csrc/conv3d_wino2x.hip (and its -DESTD_W2XABL=<mask> timing ablations): no torch, no library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iestdepth_amd/csrc [-DESTD_W2XABL=k] tools/src/w2x_time.hip -o tools/bin/w2x_time_k
//   tools/bin/w2x_time_k [N] [iters] [rbk: 0 plain, 1 accumulate, 2 residual, 3 both, 4 statistics]
#include "../../estdepth_amd/csrc/conv3d_wino2x.hip"
#include <cstdio>
#include <vector>
extern "C" int estd_get_reserved_cus(void) { return 0; }
int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 3, iters = argc > 2 ? atoi(argv[2]) : 40, mode = argc > 3 ? atoi(argv[3]) : 0;
    const int D = 64, H = 120, W = 160;
    const size_t vox = (size_t)N * D * H * W;
    float *x, *y, *r, *w, *ss;
    double* st;
    (void)hipMalloc(&x, vox * 32 * 4); (void)hipMalloc(&y, vox * 32 * 4); (void)hipMalloc(&r, vox * 32 * 4);
    (void)hipMalloc(&w, 48 * 4096); (void)hipMalloc(&ss, 64 * 4); (void)hipMalloc(&st, (size_t)N * D * 15 * 10 * 4 * 8);
    std::vector<float> hx(vox * 32);
    unsigned s = 12345u;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9) % 2001 - 1000) * 1e-3f; }
    (void)hipMemcpy(x, hx.data(), vox * 32 * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(r, hx.data(), vox * 32 * 4, hipMemcpyHostToDevice);
    std::vector<float> hw(48 * 1024);
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9) % 2001 - 1000) * 5e-5f; }
    (void)hipMemcpy(w, hw.data(), 48 * 4096, hipMemcpyHostToDevice);
    std::vector<float> hs(64, 1.0f);
    for (int i = 32; i < 64; ++i) hs[i] = 0.01f;
    (void)hipMemcpy(ss, hs.data(), 256, hipMemcpyHostToDevice);
    estd_conv3d_desc d{};
    d.N = N; d.D = D; d.H = H; d.W = W; d.cin_main = 32; d.in_stride = 32; d.n_tiles = 2;
    d.in_main = x; d.w_wino2 = w; d.scale = ss; d.shift = ss + 32; d.act_a = d.act_b = ESTD_ACT_RELU;
    d.out_main = y; d.out_stride = 32; d.out_channels = 32; d.out_scale = 1.0f;
    if (mode == 1 || mode == 3) d.accumulate = 1;
    if (mode == 2 || mode == 3) d.residual = r;
    if (mode == 4) d.stats_partials = st;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 400; ++i) if (estd_conv3d_k3_wino2x(&d, nullptr) != 0) { printf("launch failed\n"); return 1; }
    (void)hipDeviceSynchronize();
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) (void)estd_conv3d_k3_wino2x(&d, nullptr);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= iters; sum += ms; if (ms < best) best = ms;
    }
    printf("ABL=%d N=%d mode=%d: %.4f ms (best %.4f)\n", ESTD_W2XABL, N, mode, sum / 3, best);
    return 0;
}
