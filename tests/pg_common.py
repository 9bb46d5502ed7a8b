"""Shared small proving configuration for the L1 (txn/agg/block) tests."""
SMALL = dict(
    table_log_lo=[6, 5, 6, 7, 5, 6, 8], table_log_hi=[8, 7, 8, 9, 7, 8, 10],
    stark_rate_bits=1, stark_cap_height=4, stark_num_queries=10, stark_pow_bits=8, arity_bits=4, final_poly_bits=5,
    rec_log_n=6, rec_n_cols=19, rec_n_const=5, rec_rate_bits=3, rec_num_queries=6, rec_pow_bits=6, shrink_depth=2,
    rec_air_id=0)   # recursion-shaped proofs on the synthetic AIR (small: 19 columns); SMALL_PLONK below is the default kind
LOG_N = (6, 5, 7, 7, 5, 6, 9)
WIDTH = (16, 8, 24, 40, 16, 24, 8)
IR_MAGIC = 0x52494E5854475042


def ir_words(block_number, txn_before, seed, root_before=(1, 2, 3, 4), gas=(100, 121), log_n=LOG_N, width=WIDTH):
    return [IR_MAGIC, 1, block_number, txn_before, gas[0], gas[1], *root_before, seed, *log_n, *width]

# the same with the recursion-shaped proofs made on the PLONK-shaped circuit (AIR 8: 135 wires, 85 preprocessed constants)
SMALL_PLONK = dict(SMALL, rec_n_cols=135, rec_n_const=85, rec_air_id=8)
