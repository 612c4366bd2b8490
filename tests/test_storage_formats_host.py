"""convert_to_format never changes a product's bits (missing 6).

SparseMatrix::from_triplets always builds CSR (matrix/mod.rs:160-199); COO, CSC and Graph storage exist only as convert_to_format() of
it (matrix/mod.rs:244-296), each filled from to_triplets() of the storage before.  Every chain of conversions therefore hands the
multiply loop of the new storage (sparse.rs:409-430 CSC, :584-597 COO, :763-773 Graph) a row's entries in ascending column order with
duplicates in their original insertion order — the sequence CSRStorage::multiply_vector adds them in (sparse.rs:187-203).  Shown here on
the oracle's restatements of the four loops, with duplicates, signed zeros in x and every conversion path; the consequence for the
device: sl_spmv / sl_spmv_add answer for a SparseMatrix in ANY storage format, and convert_to_format needs no device work."""
import numpy as np

from oracle import oracle as O


def _system(n, seed, dup=True):
    rng = np.random.default_rng(seed)
    nnz = 9 * n
    r, c = rng.integers(0, n, nnz), rng.integers(0, n, nnz)
    v = rng.standard_normal(nnz) * 10.0 ** rng.integers(-6, 7, nnz)
    if dup:      # the same (row, col) several times, in an insertion order that matters for the sum
        r[: n // 2], c[: n // 2] = r[n: n + n // 2], c[n: n + n // 2]
    x = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4, n)
    x[rng.integers(0, n, n // 10)] = 0.0
    x[rng.integers(0, n, n // 20)] = -0.0
    return r, c, v, x


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_every_storage_reachable_through_sparse_matrix_multiplies_to_the_same_bits():
    for n, seed in ((50, 1), (400, 2), (1000, 3)):
        r, c, v, x = _system(n, seed)
        rp, ci, va = O.csr_from_triplets(r, c, v, n, n)                       # SparseMatrix::from_triplets -> CSR (stable (row, col) sort)
        y_csr = O.spmv(rp, ci, va, x)
        rows_csr = np.repeat(np.arange(n, dtype=np.uint32), np.diff(rp))       # CSRStorage::to_triplets: row-major, stored order
        # CSR -> COO (via to_triplets)
        assert (_bits(O.spmv_coo(n, rows_csr, ci, va, x)) == _bits(y_csr)).all()
        # CSR -> CSC (CSCStorage::from_csr = from_coo of the CSR's triplets)
        cp, ro, vo = O.coo_to_csc(n, rows_csr, ci, va)
        assert (_bits(O.spmv_csc(n, cp, ro, vo, x)) == _bits(y_csr)).all()
        # CSC -> COO: the triplets now come column-major (CSCStorage::to_triplets) — a row's products still arrive in ascending column order
        cols_csc = np.repeat(np.arange(n, dtype=np.uint32), np.diff(cp))
        assert (_bits(O.spmv_coo(n, ro, cols_csc, vo, x)) == _bits(y_csr)).all()
        # -> Graph from either order of triplets
        assert (_bits(O.spmv_graph(n, rows_csr, ci, va, x)) == _bits(y_csr)).all()
        assert (_bits(O.spmv_graph(n, ro, cols_csc, vo, x)) == _bits(y_csr)).all()
        # CSC -> CSR -> the same arrays again (CSRStorage::from_csc = from_coo of column-major triplets: stable (row, col) sort)
        rp2, ci2, va2 = O.csr_from_triplets(ro, cols_csc, vo, n, n)
        assert (rp2 == rp).all() and (ci2 == ci).all() and (_bits(va2) == _bits(va)).all()


def test_the_insertion_order_of_raw_triplets_is_what_would_differ():
    """what the review had in mind: a COOStorage filled from RAW triplets adds in insertion order — different bits; no SparseMatrix holds one"""
    r, c, v, x = _system(400, 7)
    rp, ci, va = O.csr_from_triplets(r, c, v, 400, 400)
    keep = v != 0.0
    y_raw = O.spmv_coo(400, r[keep], c[keep], v[keep], x)
    y_csr = O.spmv(rp, ci, va, x)
    assert (_bits(y_raw) != _bits(y_csr)).any() and np.allclose(y_raw, y_csr, rtol=1e-9, atol=1e-6 * np.abs(y_csr).max())
