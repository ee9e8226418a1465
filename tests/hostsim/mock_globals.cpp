// TEST INFRASTRUCTURE ONLY: fiber scheduler + wave collectives of the mock HIP runtime (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>
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
#if !defined(__x86_64__)
#error "the mock runtime's context switch is written for x86-64"
#endif
extern "C" void mock_ctx_switch(void **save_sp, void *load_sp);
asm(".text\n.globl mock_ctx_switch\n.type mock_ctx_switch,@function\nmock_ctx_switch:\n"
	"  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
	"  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
	"  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
	".size mock_ctx_switch, .-mock_ctx_switch\n");
namespace {
const size_t STACK = 512 * 1024;
struct Wave { uint64_t buf[2][64]; uint64_t mask[2]; uint64_t live_mask = 0; int live = 0, arrived = 0; unsigned gen = 0; };
// Context switches are hand-made (callee-saved registers + stack pointer): glibc's swapcontext saves and restores the signal mask with
// two system calls per switch, and a wave collective costs every lane two switches -- a third of the suite's time went there.
struct Fiber { void *sp = nullptr; bool done = false; dim3 tid; char *stack = nullptr; };
std::vector<Fiber> fibers;
std::vector<Wave> waves;
std::vector<char*> stacks;
void *sched_sp = nullptr;
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
	mock_ctx_switch(&f.sp, sched_sp);
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
		mock_ctx_switch(&fibers[me].sp, sched_sp);
	}
	memcpy(out, w.buf[g & 1], sizeof(uint64_t) * 64);
	*active_mask = w.mask[g & 1];
}

void mock_block_barrier()
{
	unsigned g = blk_gen;
	if (++blk_arrived == blk_live) { blk_arrived = 0; ++blk_gen; }
	while (blk_gen == g) { int me = cur; mock_ctx_switch(&fibers[me].sp, sched_sp); }
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
			// a fresh fiber's stack: six zeroed callee-saved registers, then fiber_main as the switch's return address (entered with the
			// stack pointer 8 below a 16-byte boundary, as after a call), then a null return address fiber_main never uses
			void **top = (void**)(((uintptr_t)stacks[t] + STACK) & ~(uintptr_t)15);
			top[-1] = nullptr; top[-2] = (void*)fiber_main;
			for (int k = 3; k <= 8; ++k) top[-k] = nullptr;
			f.sp = (void*)(top - 8);
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
				mock_ctx_switch(&sched_sp, fibers[t].sp);
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
