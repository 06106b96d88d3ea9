/*
 * homulator_host.h — C entry points of the host layer (libhomulator_host.so): build one FHE operation with
 * the Operation / InsGen / Driver classes and execute it on the backend.  Used by bench.py and the parity
 * tests through ctypes; the CLI (host/bench_test/bench_micro24.cpp) uses the C++ classes directly.
 * Mirrors the reference's only caller, main() of bench_test/bench_micro24.cpp:5-52.
 */
#ifndef HOMULATOR_HOST_H
#define HOMULATOR_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct hh_op hh_op;

/* op in {hmult, hrotate, hadd, pmult, padd}; backend: 0 = hip, 1 = count (no GPU), 2 = sim (no GPU: the cycle model of
 * the reference accelerator, host/include/SimModel.h); fuse: 0/1;
 * extra "key=value" config overrides separated by ';' (may be NULL), e.g. "galois=25;seed=7".
 * quiet != 0 suppresses the constructors' stdout (config echo, Malloc lines). */
int hh_op_create(hh_op **op, const char *cfg_path, const char *op_name, uint32_t max_level, uint32_t cur_level,
                 uint32_t alpha, int backend, int fuse, int device, const char *overrides, int quiet);
void hh_op_destroy(hh_op *op);
const char *hh_last_error(void);

int hh_op_simulate(hh_op *op);                              /* upstream entry: prints banner + stat block */
int hh_op_execute(hh_op *op, uint32_t iters, double *ns_per_iter); /* whole op, device time */
/* backend = sim: runs the cycle model to completion (silently).  cycles = the reference's "FHE-Sim Total simulated"; retired =
 * instructions written back; drained = 0 if upstream's dead-lock exit was taken (no instruction retired for 2000 cycles).
 * hh_op_sim_stats: the reference's stat block as "key value\n" lines, in its order. */
int hh_op_sim_run(hh_op *op, uint64_t *cycles, uint64_t *retired, int *drained);
int hh_op_sim_stats(hh_op *op, char *out, uint32_t cap);
int hh_op_enqueue(hh_op *op, uint32_t iters);               /* asynchronous: no timing, no sync */
int hh_op_sync(hh_op *op);
int hh_op_total_instructions(hh_op *op, uint64_t *total);
int hh_op_launch_count(hh_op *op, uint64_t *n);
int hh_op_stage_bytes(hh_op *op, uint64_t *bytes);          /* sum over launches of operand bytes */
int hh_op_buffer_limbs(hh_op *op, const char *name, uint32_t *n_limbs);
int hh_op_read_buffer(hh_op *op, const char *name, uint64_t *host); /* [n_limbs][N] */
int hh_op_buffer_names(hh_op *op, char *out, uint32_t cap); /* '\n'-separated */
uint32_t hh_op_N(hh_op *op);
/* overrides "batch=B": every launch carries B independent ops (own inputs: fill seeds + copy * 100000; shared evaluation key) */
int hh_op_read_buffer_copy(hh_op *op, const char *name, uint32_t copy, uint64_t *host);
/* real data in: overwrite a named input / key buffer ([n_limbs][N] words, fully reduced, evaluation form) of op `copy` of the batch; the
 * evaluation-key limbs ("IP_Key<k>_<j>", shared by the batch) live in copy 0.  Prepares the op if needed; synchronises. */
int hh_op_write_buffer(hh_op *op, const char *name, uint32_t copy, const uint64_t *host);
uint32_t hh_op_batch(hh_op *op);
int hh_op_plan(hh_op *op, char *out, uint32_t cap);          /* launch plan, one line per launch */
/* hm_get_counter of the op's backend context (hip backend; e.g. "ntt_cross_xcd", "ntt_fused_small", "arith"): synchronises */
int hh_op_backend_counter(hh_op *op, const char *name, uint64_t *value);
int hh_op_stage_times(hh_op *op, uint32_t iters, char *out, uint32_t cap); /* "<kind> <stages> <ns>" per launch, each timed alone */
/* continuous execution: `dst`'s input ciphertext `input` ("ct1" / "ct2") becomes `src`'s output ciphertext,
 * copied device to device before every run of `dst`, stream-ordered after `src`; call before the first execute of `dst` and after `src` was
 * executed or prepared.  A chain builds the ops itself: op_list = "hmult,hrotate,hadd,...", levels follow the data. */
int hh_op_bind_input(hh_op *dst, const char *input, hh_op *src);
typedef struct hh_chain hh_chain;
int hh_chain_create(hh_chain **out, const char *cfg_path, const char *op_list, uint32_t maxLevel, uint32_t curLevel, uint32_t alpha,
                    const char *overrides, int quiet);
void hh_chain_destroy(hh_chain *c);
uint32_t hh_chain_size(hh_chain *c);
hh_op *hh_chain_op(hh_chain *c, uint32_t i);   /* borrowed handle of op i (read buffers, plan); do not destroy */
int hh_chain_execute(hh_chain *c, uint32_t iters, double *ns_per_pass);
int hh_chain_enqueue(hh_chain *c, uint32_t iters);            /* asynchronous passes over the chain */
int hh_chain_sync(hh_chain *c);
/* stream-ordered helpers (asynchronous): new synthetic data for an input ("ct1", "ct2": .c0 gets `seed`, .c1 `seed` + 1000,
 * as at construction); device-side snapshot of a named buffer into `slot`; download of a slot (synchronises) */
int hh_op_refill(hh_op *op, const char *input, uint64_t seed);
int hh_op_snapshot(hh_op *op, const char *name, uint32_t slot);
int hh_op_snapshot_read(hh_op *op, uint32_t slot, uint64_t *host);
int hh_chain_simulate(hh_chain *c);
/* multi-GPU (overrides "world=W;rank=R" at creation): set the transport before the first execute */
int hh_comm_unique_id(void *out128);
int hh_op_comm_init_rccl(hh_op *op, const void *unique_id128);
int hh_op_comm_init_external(hh_op *op, void *exchange_fn, void *user); /* hm_exchange_fn of homulator_hip.h */
/* the CLI's rendezvous file for the RCCL unique id (host/include/RcclRendezvous.h), exposed for its tests: the run-unique path for
 * the current environment, publish (atomic), fetch (waits up to timeout_ms), remove */
int hh_rccl_id_path(char *out, uint32_t cap);
int hh_rccl_id_publish(const char *path, const void *id128);
int hh_rccl_id_fetch(const char *path, void *id128, uint32_t timeout_ms);
int hh_rccl_id_remove(const char *path);
#ifdef __cplusplus
}
#endif
#endif
