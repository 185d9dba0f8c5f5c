/* libtsasr_lab.so - LAB EQUIPMENT, not part of the product ABI (include/tsasr_hip.h): three aids that tests / tools use to stress or
 * time the product library from the outside. Built by `make -C ts-asr_amd/csrc` next to the product library, loaded only by
 * tests/helpers/, tools/ and ts-asr_amd/prof.py's TSASR_STAMPS mode (ts-asr_amd/_capi.py::lab()). Nothing here computes anything of the
 * training step. */
#ifndef TSASR_LAB_H
#define TSASR_LAB_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
/* overwrite the whole LDS of every CU with a 32-bit pattern: a kernel that reads LDS it never wrote then sees the pattern, not leftovers
 * (tests/helpers/lds_garbage.py) */
int tsasr_lab_fill_lds(unsigned pattern, void *stream);
/* nwords 32-bit words at p <- pattern (poisoning the captured step's free pool memory between replays: tools/det_stress.py) */
int tsasr_lab_fill(void *p, unsigned pattern, size_t nwords, void *stream);
/* *out (uint64, device) = the device wall clock (100 MHz ticks) when a one-thread kernel reaches the head of `stream`
 * (tools/step_stamps.py: phase stamps inside an unprofiled replay of the captured step) */
int tsasr_lab_stamp(void *out, void *stream);
#ifdef __cplusplus
}
#endif
#endif
