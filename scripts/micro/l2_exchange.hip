// XCD-scoped operand exchange through L2 (round 6, VERDICT r05 "Next round" item 2, step (a): the gate before any kernel work).
//
// Question: can the 32 workgroups of an XCD hand bf16 operand panels to each other through the XCD's 4-MiB L2 instead of HBM?
// The encoder backward writes ~290 MB of weight-gradient operands per block that wgrad_dma reads back one launch later; a CU
// cannot hold dW (540 KB), but the 32 CUs of an XCD could hold 1/32 each if the operands they need were L2-resident.
//
// Protocol measured here (256 persistent workgroups, one per CU, group = blockIdx.x % 8 as the library's tile maps assume;
// the XCC id each workgroup really ran on is read from HW_REG_XCC_ID and the grouping is checked on the host):
//   per iteration: every workgroup writes its panel (P bytes, 16-byte plain stores) into half (it & 1) of its group's ring,
//   every storing wave drains vmcnt, one lane adds to the group's counter (agent-scope atomic; mode R: an agent-scope release
//   fence in front of it = the placement-independent form), one lane polls the counter (sc1 loads + s_sleep) until all 32
//   have arrived, then every workgroup reads `fr`/32 of EVERY panel of the half with sc1 loads (L1 bypassed: the lines were
//   rewritten by other CUs) and checks every word.  Two halves + one barrier per iteration are WAR-safe: a workgroup reaches
//   barrier it + 1 only after its reads of iteration it, and nobody writes half (it & 1) again before barrier it + 1 is complete.
//   100 iterations over the SAME ring.
// Reported: us per iteration for write only / write + barrier / write + barrier + read, the barrier's own cost (empty panels),
// words that read stale, and — from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same binary (scripts/gpu_r06_a.sh) —
// the fabric-side bytes against ring bytes x iterations.
// Gate (VERDICT): HBM bytes << ring bytes x iterations and <= 3 us per barrier.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/l2_exchange.hip -o scripts/micro/l2_exchange.bin
//   l2_exchange.bin [mode: 0 all | 1 write | 2 write+barrier | 3 full]   (modes 1-3: one configuration, for the PMC passes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(1))) unsigned int gu32;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NT = 256, GROUPS = 8, MEMBERS = 32;


__device__ __forceinline__ unsigned word_of(int it, int wg, int idx) { return (unsigned)(it * 2654435761u) ^ (unsigned)(wg << 20) ^ (unsigned)idx; }

// WRITE: panels are written; BAR: the group barrier runs; FR: 32nds of every panel each workgroup reads back (0: none);
// REL: agent-scope release fence (buffer_wbl2) in front of the arrival = placement-independent form
template <bool WRITE, bool BAR, int FR, bool REL, bool CHECK = true>
__global__ __launch_bounds__(NT) void exch(unsigned char* ring, unsigned* counters, int iters, int panel_bytes, unsigned* stale,
                                           unsigned* xcc_out, unsigned* tmo) {
    const int b = blockIdx.x, grp = b % GROUPS, mem = b / GROUPS;
    if (threadIdx.x == 0) xcc_out[b] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;       // HW_REG_XCC_ID[3:0]
    const size_t half_bytes = (size_t)MEMBERS * panel_bytes;
    unsigned char* gring = ring + (size_t)grp * 2 * half_bytes;
    gu32* cnt = (gu32*)(counters + grp * 64);                                                  // one 256-byte line per group
    const int vec_per_thread = panel_bytes / 16 / NT;                                          // 16-byte stores per thread and panel
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned char* half = gring + (size_t)(it & 1) * half_bytes;
        if (WRITE) {
            u32x4* dst = reinterpret_cast<u32x4*>(half + (size_t)mem * panel_bytes);
            for (int v = 0; v < vec_per_thread; ++v) {
                const int i = v * NT + threadIdx.x;
                u32x4 w;
                for (int e = 0; e < 4; ++e) w[e] = word_of(it, b, i * 4 + e);
                dst[i] = w;
            }
        }
        if (BAR) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   // every storing wave drains
            __syncthreads();
            if (threadIdx.x == 0) {
                if (REL) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = (unsigned)MEMBERS * (it + 1);
                unsigned spins = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 22)) { atomicAdd(tmo, 1u); break; }                   // bounded: a hang would be a gpurun strike
                }
            }
            __syncthreads();
        }
        if (FR > 0) {
            // slice `mem` (and the FR - 1 following ones, cyclically) of every panel of the half, sc1 loads (aux 16): L1 bypassed
            const int slice_bytes = panel_bytes / MEMBERS;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(half, 0, (int)half_bytes, 0x00020000);
            const int vec_per_slice = slice_bytes / 16;
            const int total = MEMBERS * FR * vec_per_slice;
            for (int i0 = threadIdx.x; i0 < total; i0 += 8 * NT) {              // 8 loads in flight per lane
                u32x4 w[8]; int vi[8], wg[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * NT;
                    const int p = i / (FR * vec_per_slice), r = i % (FR * vec_per_slice);
                    const int sl = (mem + r / vec_per_slice) % MEMBERS;
                    vi[u] = sl * vec_per_slice + r % vec_per_slice; wg[u] = p * GROUPS + grp;
                    w[u] = i < total ? __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, p * panel_bytes + vi[u] * 16, 0, 16)) : u32x4{0, 0, 0, 0};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + u * NT < total)
                        for (int e = 0; e < 4; ++e) bad += !CHECK ? (w[u][e] == 0x9e3779b9u) : (WRITE && BAR) ? (w[u][e] != word_of(it, wg[u], vi[u] * 4 + e)) : (w[u][e] == 0x9e3779b9u);
            }
        }
    }
    if (bad) atomicAdd(stale, bad);
}

struct Res { float us_iter; unsigned stale, tmo; bool grouped; };

template <bool WRITE, bool BAR, int FR, bool REL, bool CHECK = true>
Res run(unsigned char* ring, unsigned* counters, int iters, int panel_bytes, unsigned* dflags, unsigned* xcc) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    // warm-up launch, then the timed one; counters zeroed before each (epochs count within a launch)
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemsetAsync(counters, 0, GROUPS * 256, 0));
        CHECK(hipMemsetAsync(dflags, 0, 16, 0));
        if (rep == 1) CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((exch<WRITE, BAR, FR, REL, CHECK>), dim3(256), dim3(NT), 0, 0, ring, counters, iters, panel_bytes, dflags, xcc, dflags + 1);
        if (rep == 1) CHECK(hipEventRecord(b, 0));
    }
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    unsigned h[4]; CHECK(hipMemcpy(h, dflags, 16, hipMemcpyDeviceToHost));
    std::vector<unsigned> hx(256); CHECK(hipMemcpy(hx.data(), xcc, 1024, hipMemcpyDeviceToHost));
    bool grouped = true;
    for (int i = 0; i < 256; ++i) grouped &= hx[i] == hx[i % GROUPS];
    return Res{ms * 1e3f / iters, h[0], h[1], grouped};
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int iters = 100;
    unsigned char* ring; unsigned *counters, *dflags, *xcc;
    CHECK(hipMalloc(&ring, (size_t)GROUPS * 2 * MEMBERS * (128 << 10)));
    CHECK(hipMemset(ring, 0, (size_t)GROUPS * 2 * MEMBERS * (128 << 10)));
    CHECK(hipMalloc(&counters, GROUPS * 256)); CHECK(hipMalloc(&dflags, 16)); CHECK(hipMalloc(&xcc, 1024));
    if (mode != 0) {          // one configuration for the PMC passes: panel 32 KB = 1-MB halves, 2-MB ring per XCD (modes 1-5);
                              // modes 6 / 7: panel 128 KB (one 48-row panel of enc_mlp_bwd per CU: 4-MB halves), write + barrier / + read 8/32
        const int pb = mode >= 6 ? (128 << 10) : (32 << 10);
        Res r = mode == 1 ? run<true, false, 0, false>(ring, counters, iters, pb, dflags, xcc)
              : mode == 2 ? run<true, true, 0, false>(ring, counters, iters, pb, dflags, xcc)
              : mode == 3 ? run<true, true, 1, false>(ring, counters, iters, pb, dflags, xcc)
              : mode == 4 ? run<true, true, 8, false>(ring, counters, iters, pb, dflags, xcc)
              : mode == 5 ? run<true, true, 8, true>(ring, counters, iters, pb, dflags, xcc)
              : mode == 6 ? run<true, true, 0, false>(ring, counters, iters, pb, dflags, xcc)
                          : run<true, true, 8, false>(ring, counters, iters, pb, dflags, xcc);
        printf("mode %d panel %d KB: %.2f us/iter, ring bytes x iterations = %.1f MB written, stale %u, timeouts %u, grouped %d\n", mode, pb >> 10,
               r.us_iter, 256.0 * pb * iters / 1e6, r.stale, r.tmo, (int)r.grouped);
        return 0;
    }
    {
        Res r = run<false, true, 0, false>(ring, counters, iters, 16 << 10, dflags, xcc);
        printf("barrier alone (no payload, atomic arrive + sc1 poll, 32 workgroups per group): %.2f us per barrier, timeouts %u, workgroups b and b+8 on one XCC: %s\n",
               r.us_iter, r.tmo, r.grouped ? "yes" : "NO");
        r = run<false, true, 0, true>(ring, counters, iters, 16 << 10, dflags, xcc);
        printf("barrier alone with an agent release fence before the arrival: %.2f us per barrier\n", r.us_iter);
    }
    for (int kb : {8, 16, 32, 64, 128}) {       // 128 KB = the operands of ONE 48-row panel of enc_mlp_bwd (126 KB): 8 MB per XCD for the two halves
        const int pb = kb << 10;
        const double ringmb = 2.0 * MEMBERS * pb / 1048576.0;
        Res w = run<true, false, 0, false>(ring, counters, iters, pb, dflags, xcc);
        Res wb = run<true, true, 0, false>(ring, counters, iters, pb, dflags, xcc);
        Res f1 = run<true, true, 1, false>(ring, counters, iters, pb, dflags, xcc);
        Res f8 = run<true, true, 8, false>(ring, counters, iters, pb, dflags, xcc);
        Res f32 = run<true, true, 32, false>(ring, counters, iters, pb, dflags, xcc);
        Res r8 = run<true, true, 8, true>(ring, counters, iters, pb, dflags, xcc);
        printf("panel %3d KB (ring %4.1f MB per XCD, %5.1f MB written chip-wide per iteration):\n", kb, ringmb, 256.0 * pb / 1e6);
        printf("   write only                      %7.2f us/iter  (%5.2f TB/s of stores)\n", w.us_iter, 256.0 * pb / w.us_iter / 1e6);
        printf("   write + barrier                 %7.2f us/iter\n", wb.us_iter);
        printf("   + read 1/32 of every panel      %7.2f us/iter  stale words %u  timeouts %u\n", f1.us_iter, f1.stale, f1.tmo);
        printf("   + read 8/32 of every panel      %7.2f us/iter  stale words %u  (L2 -> CU %5.2f TB/s)\n", f8.us_iter, f8.stale, 256.0 * 8 * pb / f8.us_iter / 1e6);
        printf("   + read ALL of every panel       %7.2f us/iter  stale words %u  (L2 -> CU %5.2f TB/s)\n", f32.us_iter, f32.stale, 256.0 * 32 * pb / f32.us_iter / 1e6);
        printf("   8/32 with release fence         %7.2f us/iter  stale words %u\n", r8.us_iter, r8.stale);
        Res n8 = run<true, true, 8, false, false>(ring, counters, iters, pb, dflags, xcc);
        Res n32 = run<true, true, 32, false, false>(ring, counters, iters, pb, dflags, xcc);
        printf("   8/32, words folded not checked  %7.2f us/iter  (L2 -> CU %5.2f TB/s);  ALL: %7.2f us/iter (%5.2f TB/s)\n", n8.us_iter, 256.0 * 8 * pb / n8.us_iter / 1e6,
               n32.us_iter, 256.0 * 32 * pb / n32.us_iter / 1e6);
    }
    return 0;
}
