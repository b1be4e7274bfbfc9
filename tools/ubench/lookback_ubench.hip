// lookback_ubench.hip -- what the synchronisation of a ONE-SWEEP radix pass would cost on MI355X at this sort's size, measured.
// A one-sweep (decoupled look-back) pass replaces the per-pass histogram kernel + row-scan kernel (9.5 + 5 us at 3 M keys) by a
// look-back inside the scatter kernel: every block publishes its 256 digit counts, then obtains the sum over all EARLIER blocks by
// walking back over their published (partial or inclusive) status words.  This kernel does exactly that part, with the sort's geometry --
// 730 blocks (4 096 keys each) x 256 threads, one thread per digit, dynamic block ids from a ticket -- and nothing else; all blocks are
// co-resident on the chip (2 920 waves), as they are in the real pass.
//   hipcc --offload-arch=gfx950 -O3 -o lookback_ubench lookback_ubench.hip && ./lookback_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kDigits = 256;
constexpr unsigned kPartial = 1u << 30, kInclusive = 2u << 30, kMask = (1u << 30) - 1u;

// mode 0: publish only (the floor); 1: classic look-back, one thread per digit walks back one block at a time;
// 2: wave-parallel look-back: the 64 lanes of a wave read 64 predecessors of ONE digit at a time (4 waves x 64 digits each)
// ORDER (round 6): the round-4 version published and polled with RELEASE / ACQUIRE at agent scope.  On a chip of eight XCDs with one L2
// each that is a cache write-back per store and an L2 INVALIDATE per poll -- 256 of them per block and step: 86 / 770 / 2 300 us said
// nothing about look-back and everything about those cache operations.  A status word carries flag AND value, so no other memory has to
// be ordered against it: RELAXED atomics at agent scope (stores and loads that go to the memory side of the L2s, no write-back, no
// invalidate) are all the algorithm needs, and a polling thread sleeps between attempts instead of hammering the fabric.
#ifndef LB_ORDER_STORE
#define LB_ORDER_STORE __ATOMIC_RELAXED
#define LB_ORDER_LOAD __ATOMIC_RELAXED
#endif
template <int MODE>
__global__ __launch_bounds__(kDigits) void lookback_kernel(unsigned* __restrict__ status, unsigned* __restrict__ ticket, unsigned* __restrict__ out, unsigned work) {
    __shared__ unsigned s_id;
    if (threadIdx.x == 0) s_id = atomicAdd(ticket, 1u);
    __syncthreads();
    const unsigned b = s_id, d = threadIdx.x;
    // stand-in for the block's own work before it knows its counts (loading 4 096 keys and counting digits takes a few microseconds)
    unsigned count = 1u + ((b * 2654435761u + d * 40503u) >> 28);
    for (unsigned i = 0; i < work; ++i) count = (count * 1664525u + 1013904223u) & 15u | 1u;
    unsigned* mine = status + (size_t)b * kDigits;
    __hip_atomic_store(&mine[d], (b == 0 ? kInclusive : kPartial) | count, LB_ORDER_STORE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned excl = 0;
    if (MODE == 1 && b > 0) {
        for (int p = (int)b - 1; p >= 0; --p) {
            unsigned v;
            while (((v = __hip_atomic_load(&status[(size_t)p * kDigits + d], LB_ORDER_LOAD, __HIP_MEMORY_SCOPE_AGENT)) >> 30) == 0u) __builtin_amdgcn_s_sleep(1);
            excl += v & kMask;
            if ((v >> 30) == 2u) break;
        }
        __hip_atomic_store(&mine[d], kInclusive | ((excl + count) & kMask), LB_ORDER_STORE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (MODE == 2 && b > 0) {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        __shared__ unsigned s_excl[kDigits];
        for (int dd = w * 64; dd < w * 64 + 64; ++dd) {      // this wave's 64 digits, one after the other; 64 predecessors per step
            unsigned sum = 0;
            int hi = (int)b - 1;
            bool done = false;
            while (!done && hi >= 0) {
                const int p = hi - lane;
                unsigned v = kInclusive;                        // lanes past block 0 behave like an inclusive zero
                if (p >= 0) { while (((v = __hip_atomic_load(&status[(size_t)p * kDigits + dd], LB_ORDER_LOAD, __HIP_MEMORY_SCOPE_AGENT)) >> 30) == 0u) __builtin_amdgcn_s_sleep(1); }
                const unsigned long long inc = __builtin_amdgcn_ballot_w64((v >> 30) == 2u);
                const int first_inc = inc ? __builtin_ctzll(inc) : 64;   // nearest inclusive predecessor among these 64
                unsigned x = lane <= first_inc ? (v & kMask) : 0u;
                for (int m = 32; m > 0; m >>= 1) x += __shfl_xor((int)x, m);
                sum += x;
                done = inc != 0ull;
                hi -= 64;
            }
            if (lane == 0) s_excl[dd] = sum;
        }
        __syncthreads();
        excl = s_excl[d];
        __hip_atomic_store(&mine[d], kInclusive | ((excl + count) & kMask), LB_ORDER_STORE, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[(size_t)b * kDigits + d] = excl;
}

template <int MODE>
static float run(int blocks, unsigned work, unsigned* status, unsigned* ticket, unsigned* out, int reps, bool check) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float total = 0.f;
    for (int r = 0; r < reps + 2; ++r) {
        hipMemsetAsync(status, 0, (size_t)blocks * kDigits * 4, 0); hipMemsetAsync(ticket, 0, 4, 0);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(lookback_kernel<MODE>, dim3(blocks), dim3(kDigits), 0, 0, status, ticket, out, work);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) total += ms;
    }
    if (check && MODE != 0) {   // exclusive prefix over the blocks in ticket order, per digit
        std::vector<unsigned> h((size_t)blocks * kDigits);
        hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
        std::vector<unsigned> run(kDigits, 0);
        bool ok = true;
        for (int b = 0; b < blocks && ok; ++b) for (int d = 0; d < kDigits; ++d) {
            unsigned c = 1u + (((unsigned)b * 2654435761u + (unsigned)d * 40503u) >> 28);
            for (unsigned i = 0; i < work; ++i) c = (c * 1664525u + 1013904223u) & 15u | 1u;
            if (h[(size_t)b * kDigits + d] != run[d]) { ok = false; printf("  MISMATCH block %d digit %d: %u != %u\n", b, d, h[(size_t)b * kDigits + d], run[d]); break; }
            run[d] += c;
        }
        printf("  prefix sums %s\n", ok ? "correct" : "WRONG");
    }
    return total / reps * 1e3f;
}

int main() {
    for (int blocks : {123, 730, 1465}) {   // 500 k, 3 M, 6 M keys at 4 096 per block
        unsigned *status, *ticket, *out;
        hipMalloc(&status, (size_t)blocks * kDigits * 4); hipMalloc(&ticket, 4); hipMalloc(&out, (size_t)blocks * kDigits * 4);
        for (unsigned work : {0u, 2000u}) {   // 2000 dependent integer ops ~ the few microseconds a block spends counting its digits
            const float t0 = run<0>(blocks, work, status, ticket, out, 20, false);
            const float t1 = run<1>(blocks, work, status, ticket, out, 20, true);
            const float t2 = run<2>(blocks, work, status, ticket, out, 20, true);
            printf("%5d blocks, %4u ops of own work: publish only %7.1f us | thread-per-digit look-back %7.1f us | wave-parallel look-back %7.1f us\n", blocks, work, t0, t1, t2);
        }
        hipFree(status); hipFree(ticket); hipFree(out);
    }
    return 0;
}
