// Semantics probe: v_cvt_pk_u8_f32 vs trunc(clamp()) ; unaligned dword / dwordx4 global loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
__global__ void k(const float *in, unsigned *out, int n) {
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= n) return;
    unsigned r = 0;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(r) : "v"(in[i]));
    out[i] = r;
}
__global__ void k2(const unsigned char *p, unsigned *out) {
    // unaligned loads at byte offsets 1,2,3 (dword) and 4 (dwordx4 at 4-byte alignment)
    int i = threadIdx.x;
    const unsigned char *q = p + 1 + i * 5;  // arbitrary alignment
    unsigned v;
    asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(q) : "memory");
    out[i] = v;
}
int main() {
    std::vector<float> h;
    for (float v : {0.0f, 0.49f, 0.5f, 0.99f, 1.0f, 1.5f, 2.5f, 3.5f, 254.5f, 254.99f, 255.0f, 255.49f, 255.5f, 256.f, 300.f, 1e9f, -0.3f, -1.f, -1e9f, NAN, INFINITY, -INFINITY, 127.99999f, 128.5f}) h.push_back(v);
    int n = h.size();
    float *d; unsigned *o; hipMalloc(&d, n * 4); hipMalloc(&o, 1024 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, n);
    std::vector<unsigned> r(n); hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
        float t = h[i]; float c = fminf(fmaxf(t, 0.f), 255.f); unsigned e = std::isnan(t) ? 255 : (unsigned)c;
        printf("cvt_pk_u8_f32(%g) = %u   trunc-clamp = %u %s\n", h[i], r[i], e, r[i] == e ? "" : "  <-- differs");
    }
    std::vector<unsigned char> b(1024); for (int i = 0; i < 1024; ++i) b[i] = (unsigned char)(i * 7 + 3);
    unsigned char *db; hipMalloc(&db, 1024); hipMemcpy(db, b.data(), 1024, hipMemcpyHostToDevice);
    k2<<<1, 64>>>(db, o);
    std::vector<unsigned> r2(64); hipMemcpy(r2.data(), o, 64 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) { unsigned e; memcpy(&e, &b[1 + i * 5], 4); if (e != r2[i]) ++bad; }
    printf("unaligned global_load_dword: %d/64 lanes wrong\n", bad);
    return 0;
}
