// TEST INFRASTRUCTURE ONLY: fiber scheduler + wave collectives of the mock HIP runtime (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>
#include <ucontext.h>
#include <stdio.h>
#include <vector>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

dim3 threadIdx, blockIdx, blockDim, gridDim;
int mock_cur_device = 0;
static unsigned char g_lds[160 * 1024] __attribute__((aligned(64)));
unsigned char *mock_dyn_lds = g_lds;

static unsigned blk_gen = 0, blk_arrived = 0, blk_live = 0;
namespace {
const size_t STACK = 512 * 1024;
struct Wave { uint64_t buf[2][64]; uint64_t mask[2]; uint64_t live_mask = 0; int live = 0, arrived = 0; unsigned gen = 0; };
struct Fiber { ucontext_t ctx; bool done = false; dim3 tid; char *stack = nullptr; };
std::vector<Fiber> fibers;
std::vector<Wave> waves;
std::vector<char*> stacks;
ucontext_t sched_ctx;
int cur = -1;
const std::function<void()> *cur_body = nullptr;

void release_if_complete(Wave &w)
{
	if (w.live > 0 && w.arrived == w.live) { w.mask[w.gen & 1] = w.live_mask; w.arrived = 0; ++w.gen; }
}
void fiber_main()
{
	(*cur_body)();
	Fiber &f = fibers[cur];
	f.done = true;
	Wave &w = waves[cur >> 6];
	--w.live; w.live_mask &= ~(1ull << (cur & 63));
	release_if_complete(w);
	--blk_live;
	if (blk_live > 0 && blk_arrived == blk_live) { blk_arrived = 0; ++blk_gen; }
	swapcontext(&f.ctx, &sched_ctx);
}
}

void mock_exchange(uint64_t v, uint64_t out[64], uint64_t *active_mask)
{
	Wave &w = waves[cur >> 6];
	unsigned g = w.gen;
	w.buf[g & 1][cur & 63] = v;
	++w.arrived;
	release_if_complete(w);
	while (w.gen == g) {           // park until the last live lane of the wave arrives
		int me = cur;
		swapcontext(&fibers[me].ctx, &sched_ctx);
	}
	memcpy(out, w.buf[g & 1], sizeof(uint64_t) * 64);
	*active_mask = w.mask[g & 1];
}

void mock_block_barrier()
{
	unsigned g = blk_gen;
	if (++blk_arrived == blk_live) { blk_arrived = 0; ++blk_gen; }
	while (blk_gen == g) { int me = cur; swapcontext(&fibers[me].ctx, &sched_ctx); }
}

void mock_launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body)
{
	if (shmem > sizeof g_lds) { fprintf(stderr, "mock: dynamic LDS too large\n"); abort(); }
	gridDim = grid; blockDim = block;
	cur_body = &body;
	unsigned nt = block.x;
	while (stacks.size() < nt) stacks.push_back((char*)malloc(STACK));
	for (unsigned bx = 0; bx < grid.x; ++bx) {
		fibers.assign(nt, Fiber());
		waves.assign((nt + 63) / 64, Wave());
		for (unsigned t = 0; t < nt; ++t) {
			Fiber &f = fibers[t];
			f.tid = dim3(t, 0, 0);
			getcontext(&f.ctx);
			f.ctx.uc_stack.ss_sp = stacks[t]; f.ctx.uc_stack.ss_size = STACK; f.ctx.uc_link = nullptr;
			makecontext(&f.ctx, (void (*)())fiber_main, 0);
			Wave &w = waves[t >> 6];
			++w.live; w.live_mask |= 1ull << (t & 63);
		}
		blk_live = nt; blk_arrived = 0;
		unsigned remaining = nt;
		while (remaining) {
			unsigned progressed = 0;
			for (unsigned t = 0; t < nt; ++t) {
				if (fibers[t].done) continue;
				cur = (int)t; blockIdx = dim3(bx, 0, 0); threadIdx = fibers[t].tid;
				swapcontext(&sched_ctx, &fibers[t].ctx);
				if (fibers[t].done) { --remaining; ++progressed; }
			}
			(void)progressed;
		}
	}
	cur = -1; cur_body = nullptr;
}

// a crash inside emulated device code should say where: print the native backtrace (build with -g; addr2line resolves it)
static void mock_segv(int sig)
{
	void *bt[64];
	int n = backtrace(bt, 64);
	const char msg[] = "mock HIP runtime: fatal signal in emulated device code; backtrace:\n";
	(void)!write(2, msg, sizeof msg - 1);
	backtrace_symbols_fd(bt, n, 2);
	_exit(128 + sig);
}
__attribute__((constructor)) static void mock_install_handlers()
{
	if (!getenv("MOCK_HIP_BACKTRACE")) return;
	static char alt[1 << 16];
	stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof alt; ss.ss_flags = 0;
	sigaltstack(&ss, nullptr);
	struct sigaction sa; memset(&sa, 0, sizeof sa);
	sa.sa_handler = mock_segv; sa.sa_flags = SA_ONSTACK;
	sigaction(SIGSEGV, &sa, nullptr); sigaction(SIGBUS, &sa, nullptr); sigaction(SIGABRT, &sa, nullptr);
}
