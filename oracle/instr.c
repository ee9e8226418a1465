/* oracle/instr.c -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/make_instr.py).
 * Counters of the instrumented reference build (oracle/_ref/bwa_instr), printed when the program exits. */
#include <stdio.h>
#include <stdlib.h>

long long orc_n_2occ4, orc_n_blk, orc_n_sa, orc_n_lf, orc_ext_calls, orc_ext_cells, orc_glb_cells, orc_w_ref;

static void orc_instr_report(void)
{
	fprintf(stderr, "[orc_instr] n_2occ4 %lld N_blk %lld N_sa %lld N_lf %lld W_ref %lld ext_calls %lld ext_cells %lld glb_cells %lld\n",
			orc_n_2occ4, orc_n_blk, orc_n_sa, orc_n_lf, orc_w_ref, orc_ext_calls, orc_ext_cells, orc_glb_cells);
}
__attribute__((constructor)) static void orc_instr_init(void) { atexit(orc_instr_report); }
