/* oracle/proofgen.c -- txn / agg / block proof composition (CPU restatement).
 * TEST INFRASTRUCTURE ONLY; "parity unpinned" by the reference (see gl.h).
 * Filled in with the L1 layer (proof_gen.rs:39-110 shapes); see DESIGN.md section 5.
 */
#include "oracle.h"
