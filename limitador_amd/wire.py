"""Wire formats shared by the C ABI (include/rl_engine.h) and its callers."""
import numpy as np

#: rl_hit — one (request x counter) record, 16 bytes.
HIT_DTYPE = np.dtype([("key", "<u8"), ("limit", "<u4"), ("delta", "<u4")], align=True)
#: rl_cell_row — a stored cell as reported by get_counters / dump_cells, 32 bytes.
CELL_ROW_DTYPE = np.dtype(
    [("key", "<u8"), ("limit", "<u4"), ("reserved", "<u4"), ("value", "<u8"), ("expiry_us", "<u8")],
    align=True,
)
#: rl_limit_row
LIMIT_ROW_DTYPE = np.dtype([("max_value", "<u8"), ("seconds", "<u8")], align=True)
#: rl_match_cond / rl_match_limit — the compiled limit table of the on-device matcher (rl_engine.h)
MATCH_COND_DTYPE = np.dtype([("key", "<u4"), ("op", "<u4"), ("value", "<u4")], align=True)
MATCH_LIMIT_DTYPE = np.dtype([("limit", "<u4"), ("ns", "<u4"), ("cond_off", "<u4"), ("n_cond", "<u4"),
                              ("n_vars", "<u4"), ("var_key", "<u4", (2,)), ("pad", "<u4")], align=True)
#: bit 31 of a limit id: counter of a limit without variables (in_memory.rs:14 `simple_limits`)
RL_SIMPLE = 0x80000000

assert HIT_DTYPE.itemsize == 16 and CELL_ROW_DTYPE.itemsize == 32 and LIMIT_ROW_DTYPE.itemsize == 16
assert MATCH_COND_DTYPE.itemsize == 12 and MATCH_LIMIT_DTYPE.itemsize == 32


def make_hits(keys, limits, deltas=1):
    """Build an rl_hit array from per-field arrays / scalars."""
    keys = np.asarray(keys, dtype=np.uint64)
    out = np.empty(keys.shape[0], dtype=HIT_DTYPE)
    out["key"] = keys
    out["limit"] = np.asarray(limits, dtype=np.uint32)
    out["delta"] = np.asarray(deltas, dtype=np.uint32)
    return out
