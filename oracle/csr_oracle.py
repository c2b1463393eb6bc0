"""CSC / CSR construction oracle.  TEST INFRASTRUCTURE ONLY.

DGL builds its destination-sorted and source-sorted index arrays lazily inside every
message-passing call the reference makes (``model/gat.py:302-312``).  The contract
restated here is a *stable* sort: edges of one segment keep ascending original edge
id, which is also what ``scipy.sparse.coo_matrix(...).tocsr()`` yields on a
duplicate-free edge list.  Integer work, compared bit for bit.
"""
import numpy as np


def csr_by_key(key: np.ndarray, other: np.ndarray, n: int):
    """Return (ptr[n+1], other_sorted[E], eid[E]) with edges stably sorted by ``key``."""
    key = np.asarray(key, dtype=np.int64)
    eid = np.argsort(key, kind="stable").astype(np.int32)
    ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(ptr, key + 1, 1)
    ptr = np.cumsum(ptr).astype(np.int32)
    return ptr, np.asarray(other)[eid].astype(np.int32), eid


def build(src: np.ndarray, dst: np.ndarray, n: int):
    """(in_ptr, in_src, in_eid, out_ptr, out_dst, out_eid), all int32."""
    in_ptr, in_src, in_eid = csr_by_key(dst, src, n)
    out_ptr, out_dst, out_eid = csr_by_key(src, dst, n)
    return in_ptr, in_src, in_eid, out_ptr, out_dst, out_eid


def segment_plan(ptr: np.ndarray, light_max: int, chunk: int):
    """Chunk list for segments longer than ``light_max`` (pure-Python loop, small cases)."""
    heavy, cptr, cnode, cbeg, cend = [], [0], [], [], []
    for v in range(len(ptr) - 1):
        b, e = int(ptr[v]), int(ptr[v + 1])
        if e - b > light_max:
            slot = len(heavy)
            heavy.append(v)
            while b < e:
                cnode.append(slot)
                cbeg.append(b)
                cend.append(min(b + chunk, e))
                b += chunk
            cptr.append(len(cnode))
    i32 = np.int32
    return (np.array(heavy, i32), np.array(cptr, i32), np.array(cnode, i32),
            np.array(cbeg, i32), np.array(cend, i32))
