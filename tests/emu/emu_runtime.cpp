// TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h): the fiber scheduler behind the host build of the kernels.
#include <sys/mman.h>
#include <ucontext.h>

#include <vector>

#include "hip/hip_runtime.h"

emu_idx3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

static int g_emu_error = 0;  // sticky until read through hipGetLastError
static char g_emu_msg[256] = "";
hipError_t hipGetLastError() { const int e = g_emu_error; g_emu_error = 0; return e; }
const char* hipGetErrorString(hipError_t) { return g_emu_msg[0] ? g_emu_msg : "emulated launch failed"; }
hipError_t hipGetDevice(int* dev) { *dev = 0; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "host-emulation");
    p->multiProcessorCount = 0;
    return hipSuccess;
}

namespace emu {

enum { READY = 0, AT_SYNC = 1, AT_WAVE = 2, DONE = 3 };
static const size_t STACK = 512 * 1024;

struct Fiber {
    ucontext_t ctx;
    int state;
    unsigned gen;      // wave operations executed so far (slot parity)
    emu_idx3 tid;
};

static std::vector<Fiber> g_fibers;
static std::vector<char*> g_stacks;
static ucontext_t g_sched;
static int g_cur = -1;
static const std::function<void()>* g_body = nullptr;
static std::vector<uint64_t> g_slots;     // [wave][parity][lane]
static std::vector<unsigned> g_slot_gen;  // [wave][parity][lane]: the wave operation (its count + 1) the slot was published for
static std::vector<unsigned char> g_pred; // [wave][parity][lane]
static std::vector<char> g_dyn;
static int g_or_acc = 0, g_or_result = 0;  // __syncthreads_or: accumulated by arrivals, published at release
static unsigned long long g_stat_dead_reads = 0, g_stat_launches = 0, g_stat_blocks = 0;

static void yield_to_scheduler() { swapcontext(&g_fibers[g_cur].ctx, &g_sched); }

// ---- LDS-DMA: immediate or deferred landing (hip/hip_runtime.h) ----------------------------------------------------------------------
struct GldsReq { void* dst; const void* src; int bytes; };
static std::vector<std::vector<GldsReq>> g_glds;  // [work-item] requests in flight
static int g_glds_mode = -1;                       // 0 immediate, 1 deferred
static unsigned long long g_glds_rand = 88172645463325252ull, g_stat_glds_unwaited = 0;
static void glds_init() {
    if (g_glds_mode >= 0) return;
    const char* e = getenv("KM_EMU_GLDS");
    g_glds_mode = (e && e[0] == 'd') ? 1 : 0;
}
void glds_issue(void* dst, const void* src, int bytes) {
    glds_init();
    if (g_glds_mode == 0) { memcpy(dst, src, (size_t)bytes); return; }
    if ((size_t)g_cur >= g_glds.size()) g_glds.resize((size_t)g_cur + 1);
    g_glds[(size_t)g_cur].push_back({dst, src, bytes});
}
void glds_wait() {
    if (g_glds_mode != 1 || (size_t)g_cur >= g_glds.size()) return;
    std::vector<GldsReq>& q = g_glds[(size_t)g_cur];
    for (size_t n = q.size(); n > 0; --n) {  // in shuffled order (the pieces of one wave do not overlap: any order is legal)
        g_glds_rand ^= g_glds_rand << 13; g_glds_rand ^= g_glds_rand >> 7; g_glds_rand ^= g_glds_rand << 17;
        const size_t k = (size_t)(g_glds_rand % n);
        memcpy(q[k].dst, q[k].src, (size_t)q[k].bytes);
        q[k] = q[n - 1];
    }
    q.clear();
}

static void fiber_entry() {
    (*g_body)();
    g_fibers[g_cur].state = DONE;
    swapcontext(&g_fibers[g_cur].ctx, &g_sched);
}

int lane_id() { return g_cur & 63; }
char* dyn_smem() { return g_dyn.data(); }

void syncthreads() {
    g_fibers[g_cur].state = AT_SYNC;
    yield_to_scheduler();
}

int syncthreads_or(int pred) {
    g_or_acc |= (pred != 0);
    g_fibers[g_cur].state = AT_SYNC;
    yield_to_scheduler();
    return g_or_result;
}

static uint64_t* slot(int wave, unsigned parity, int lane) { return &g_slots[((size_t)wave * 2 + parity) * 64 + lane]; }
static unsigned char* pred_slot(int wave, unsigned parity, int lane) { return &g_pred[((size_t)wave * 2 + parity) * 64 + lane]; }
static unsigned* slot_gen(int wave, unsigned parity, int lane) { return &g_slot_gen[((size_t)wave * 2 + parity) * 64 + lane]; }

uint64_t wave_exchange(uint64_t mine, int src, bool* ok) {
    Fiber& f = g_fibers[g_cur];
    const int wave = g_cur >> 6, lane = g_cur & 63;
    const unsigned parity = f.gen & 1u;
    *slot(wave, parity, lane) = mine;
    *slot_gen(wave, parity, lane) = f.gen + 1u;
    *pred_slot(wave, parity, lane) = 1;
    f.state = AT_WAVE;
    yield_to_scheduler();  // resumed once every live lane of this wave has published
    const unsigned mygen = f.gen + 1u;
    f.gen++;
    const int base = wave * 64, n = (int)g_fibers.size();
    if (src < 0) {
        for (int l = 0; l < 64 && base + l < n; ++l)
            if (g_fibers[base + l].state != DONE || *slot_gen(wave, parity, l) == mygen) { src = l; break; }
    }
    // (a lane that took part in THIS operation and has since run to its end - the fibers of a wave are resumed one after the other, the
    // lanes of a real wave leave together - still counts: its slot holds what it published for this operation)
    if (src < 0 || base + src >= n || (g_fibers[base + src].state == DONE && *slot_gen(wave, parity, src) != mygen)) {
        if (src != lane) g_stat_dead_reads++;
        *ok = false;
        return mine;
    }
    *ok = true;
    return *slot(wave, parity, src);
}

uint64_t wave_ballot(bool pred) {
    Fiber& f = g_fibers[g_cur];
    const int wave = g_cur >> 6, lane = g_cur & 63;
    const unsigned parity = f.gen & 1u;
    *pred_slot(wave, parity, lane) = pred ? 2 : 1;
    *slot_gen(wave, parity, lane) = f.gen + 1u;
    f.state = AT_WAVE;
    yield_to_scheduler();
    const unsigned mygen = f.gen + 1u;
    f.gen++;
    const int base = wave * 64, n = (int)g_fibers.size();
    uint64_t m = 0;
    for (int l = 0; l < 64 && base + l < n; ++l)  // (a lane that voted in this operation and has run to its end since still counts)
        if ((g_fibers[base + l].state != DONE || *slot_gen(wave, parity, l) == mygen) && *pred_slot(wave, parity, l) == 2) m |= (1ull << l);
    return m;
}

static unsigned long long g_stat_misaligned = 0;
void misaligned(const void* p, int bytes, const char* file, int line) {
    g_stat_misaligned++;
    if (!g_emu_error) snprintf(g_emu_msg, sizeof(g_emu_msg), "misaligned %d-byte vector access at %p (%s:%d)", bytes, p, file, line);
    g_emu_error = 716;  // hipErrorMisalignedAddress, reported by the launch's km_check_launch
}

static void fail(const char* what) {
    snprintf(g_emu_msg, sizeof(g_emu_msg), "%s", what);
    g_emu_error = 719;  // hipErrorLaunchFailure
}

// Order in which ready work-items are resumed within one scheduling pass.  Hardware gives no order between waves, so a
// kernel that is correct must not depend on it: the test suite runs LDS-heavy kernels under "reverse" and seeded "random"
// orders as well, which exposes a missing barrier (a consumer resumed before its producer) as a wrong result.
//   KM_EMU_SCHEDULE = forward (default) | reverse | random:<seed> | lanes:<seed>
static int g_sched_mode = -1;  // 0 forward, 1 waves reversed, 2 waves shuffled, 3 every work-item shuffled
static unsigned long long g_sched_state = 0;
static void sched_init() {
    if (g_sched_mode >= 0) return;
    const char* e = getenv("KM_EMU_SCHEDULE");
    g_sched_mode = 0;
    if (e && e[0] == 'r' && e[1] == 'e') g_sched_mode = 1;
    if (e && ((e[0] == 'r' && e[1] == 'a') || e[0] == 'l')) {
        g_sched_mode = e[0] == 'l' ? 3 : 2;
        const char* c = strchr(e, ':');
        g_sched_state = c ? strtoull(c + 1, nullptr, 10) : 1;
        if (!g_sched_state) g_sched_state = 1;
    }
}
extern "C" void emu_set_schedule(int mode, unsigned long long seed) { g_sched_mode = mode; g_sched_state = seed ? seed : 1; }
extern "C" void emu_set_glds(int deferred) { g_glds_mode = deferred ? 1 : 0; }
static unsigned sched_rand(unsigned n) {  // xorshift64*
    g_sched_state ^= g_sched_state >> 12; g_sched_state ^= g_sched_state << 25; g_sched_state ^= g_sched_state >> 27;
    return (unsigned)((g_sched_state * 2685821657736338717ull) >> 33) % n;
}

static bool run_block(unsigned nthreads) {
    sched_init();
    static std::vector<unsigned> order;
    order.resize(nthreads);
    for (unsigned t = 0; t < nthreads; ++t) {
        Fiber& f = g_fibers[t];
        f.state = READY;
        f.gen = 0;
        f.tid.x = t % blockDim.x;
        f.tid.y = (t / blockDim.x) % blockDim.y;
        f.tid.z = t / (blockDim.x * blockDim.y);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = g_stacks[t];
        f.ctx.uc_stack.ss_size = STACK;
        f.ctx.uc_link = &g_sched;
        makecontext(&f.ctx, fiber_entry, 0);
    }
    const unsigned nwaves = (nthreads + 63) / 64;
    g_or_acc = g_or_result = 0;
    std::fill(g_slot_gen.begin(), g_slot_gen.end(), 0u);
    for (auto& q : g_glds) { g_stat_glds_unwaited += q.size(); q.clear(); }  // (requests of the previous workgroup that nobody waited for: they never land)
    for (;;) {
        bool progressed = false;
        unsigned live = 0;
        {   // waves are permuted as units (their lanes run in lane order, as lockstep execution would present them); "lanes"
            // permutes every work-item, which additionally flags code that relies on lockstep within a wave
            const unsigned nw = (nthreads + 63) / 64;
            static std::vector<unsigned> worder;
            worder.resize(nw);
            for (unsigned w = 0; w < nw; ++w) worder[w] = g_sched_mode == 1 ? nw - 1 - w : w;
            if (g_sched_mode >= 2)
                for (unsigned w = nw - 1; w > 0; --w) std::swap(worder[w], worder[sched_rand(w + 1)]);
            unsigned k = 0;
            for (unsigned w = 0; w < nw; ++w)
                for (unsigned l = 0; l < 64 && worder[w] * 64 + l < nthreads; ++l) order[k++] = worder[w] * 64 + l;
            if (g_sched_mode == 3)
                for (unsigned q = nthreads - 1; q > 0; --q) std::swap(order[q], order[sched_rand(q + 1)]);
        }
        for (unsigned k = 0; k < nthreads; ++k) {
            const unsigned t = order[k];
            if (g_fibers[t].state == READY) {
                g_cur = (int)t;
                threadIdx = g_fibers[t].tid;
                swapcontext(&g_sched, &g_fibers[t].ctx);
                progressed = true;
            }
            if (g_fibers[t].state != DONE) live++;
        }
        if (live == 0) return true;
        // wave operations: complete when every live lane of the wave has arrived
        for (unsigned w = 0; w < nwaves; ++w) {
            unsigned alive = 0, waiting = 0;
            for (unsigned t = w * 64; t < std::min(nthreads, w * 64 + 64); ++t) {
                if (g_fibers[t].state != DONE) alive++;
                if (g_fibers[t].state == AT_WAVE) waiting++;
            }
            if (alive && waiting == alive) {
                for (unsigned t = w * 64; t < std::min(nthreads, w * 64 + 64); ++t)
                    if (g_fibers[t].state == AT_WAVE) g_fibers[t].state = READY;
                progressed = true;
            }
        }
        // workgroup barrier: complete when every live work-item has arrived
        unsigned at_sync = 0;
        for (unsigned t = 0; t < nthreads; ++t) at_sync += g_fibers[t].state == AT_SYNC;
        if (at_sync == live) {
            g_or_result = g_or_acc;
            g_or_acc = 0;
            for (unsigned t = 0; t < nthreads; ++t)
                if (g_fibers[t].state == AT_SYNC) g_fibers[t].state = READY;
            progressed = true;
        }
        if (!progressed) {
            fail("divergent barrier or wave operation: some live lanes never reach it");
            return false;
        }
    }
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    const unsigned nthreads = block.x * block.y * block.z;
    if (nthreads == 0 || nthreads > 1024) { fail("invalid block size"); return; }
    if (shmem > 160 * 1024) { fail("dynamic LDS request exceeds 160 KiB"); return; }
    while (g_stacks.size() < nthreads) {
        void* p = mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { fail("fiber stack allocation failed"); return; }
        g_stacks.push_back((char*)p);
    }
    g_fibers.resize(nthreads);
    const unsigned nwaves = (nthreads + 63) / 64;
    g_slots.assign((size_t)nwaves * 2 * 64, 0);
    g_slot_gen.assign((size_t)nwaves * 2 * 64, 0);
    g_pred.assign((size_t)nwaves * 2 * 64, 0);
    g_dyn.assign(shmem + 16, 0);
    g_body = &body;
    blockDim = block;
    gridDim = grid;
    g_stat_launches++;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx.x = bx; blockIdx.y = by; blockIdx.z = bz;
                g_stat_blocks++;
                if (!run_block(nthreads)) return;
            }
}

}  // namespace emu

extern "C" {
// counters for the tests: launches, workgroups, shuffle reads from lanes that had already exited, misaligned vector accesses
void emu_stats(unsigned long long* out) { out[0] = emu::g_stat_launches; out[1] = emu::g_stat_blocks; out[2] = emu::g_stat_dead_reads; out[3] = emu::g_stat_misaligned; }
}
