"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit.

Bar: row_ptr / col_idx identical; values identical bitwise (the kernels accumulate in the CPU
loop's order with separate multiply and add), which is stricter than the 1e-12 relative
tolerance the project states.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from spgemm_amd import capi, hostcsr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    c.set_option("poison_output", 1)  # an entry of C that no kernel writes shows as NaN in column 2^32 - 1
    yield c
    c.close()


HASH_MAX_NPROD_DEFAULT = 8192  # the library's default of that option


def to_ocsr(h):
    return orc.OCsr(h.rows, h.cols, h.row_ptr, h.col_idx, h.val)


def gpu_multiply(ctx, A, B):
    dA = ctx.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
    dB = dA if B is A else ctx.from_host(B.rows, B.cols, B.row_ptr, B.col_idx, B.val)
    dC = ctx.multiply(dA, dB)
    rp, ci, v = dC.to_host()
    st = ctx.stats()
    C = orc.OCsr(A.rows, B.cols, rp, ci, v)
    dC.free()
    if dB is not dA:
        dB.free()
    dA.free()
    return C, st


def check(ctx, A, B, expect_sorted=True):
    C, st = gpu_multiply(ctx, A, B)
    E, est = orc.spgemm(to_ocsr(A), to_ocsr(B), with_stats=True)
    assert st["nprod"] == est["nprod"]
    if expect_sorted:  # output rows come out sorted by column
        assert np.array_equal(C.col_idx, C.canonical().col_idx)
    orc.assert_same(C, E, rtol=0.0)
    assert st["nnz_c"] == E.nnz
    return C, st


def test_reference_fixture_test(ctx, golden_dir):
    A = hostcsr.read_mtx(f"{golden_dir}/ref_test/A.mtx")
    B = hostcsr.read_mtx(f"{golden_dir}/ref_test/B.mtx")
    C, _ = check(ctx, A, B)
    orc.assert_same(C, orc.read_mtx(f"{golden_dir}/ref_test/C_expected.mtx"), rtol=0.0)


def test_reference_fixture_first(ctx, golden_dir):
    A = hostcsr.read_mtx(f"{golden_dir}/ref_first/A.mtx")
    C, st = check(ctx, A, A)
    orc.assert_same(C, orc.read_mtx(f"{golden_dir}/ref_first/C_expected.mtx"), rtol=0.0)
    assert st["nnz_c"] == 3067


@pytest.mark.parametrize("n,per_row,band", [(3000, 9, 40), (20000, 30, 300), (5000, 3, 2500)])
def test_banded_window_path(ctx, n, per_row, band):
    A = hostcsr.banded_random(n, per_row, band, seed=n)
    _, st = check(ctx, A, A)
    assert st["rows_window"] > 0


@pytest.mark.parametrize("n", [4096, 100_000])
def test_erdos_renyi(ctx, n):
    A = hostcsr.erdos_renyi(n, 16, seed=42)
    _, st = check(ctx, A, A)
    if n > 70_000:
        assert st["rows_hash"] > 0


@pytest.mark.parametrize("len_b", [1, 4, 5, 8, 9, 16, 17])
def test_short_rows_slot_layout_boundaries(len_b):
    """Short rows at the edges of the layout that gives every A entry a fixed slot of 4, 8 or 16
    lanes (B rows of exactly 4 / 5 / 8 / 9 / 16 / 17 entries, A rows of 1 to 64 entries, so that
    entries * slot falls on either side of the 256- and 512-product limits), with duplicates and
    cancellations among the products, under both table widths and with and without the padded layouts."""
    rng = np.random.default_rng(77 + len_b)
    k, n = 3000, 200_000
    # B: every row exactly len_b entries, columns from a pool of 600 so that products collide
    pool = np.sort(rng.choice(n, 600, replace=False))
    bc = np.concatenate([np.sort(rng.choice(pool, len_b, replace=False)) for _ in range(k)])
    br = np.repeat(np.arange(k), len_b)
    bv = rng.integers(-2, 3, len(br)).astype(np.float64)
    bv[bv == 0] = 1.0
    B = hostcsr.HostCsr.from_coo(k, n, br, bc, bv).sorted_rows()
    lens = [1, 2, 4, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64] * 8
    ar = np.repeat(np.arange(len(lens)), lens)
    ac = np.concatenate([np.sort(rng.choice(k, l, replace=False)) for l in lens])
    av = rng.integers(-2, 3, len(ar)).astype(np.float64)
    av[av == 0] = 1.0
    A = hostcsr.HostCsr.from_coo(len(lens), k, ar, ac, av).sorted_rows()
    for key32 in (1, 0):
        for packed in (2, 0):
            c = capi.Context(0)
            try:
                c.set_option("window_max_span", 0)  # nothing takes the window path
                c.set_option("short_key32", key32)
                c.set_option("short_packed", packed)
                _, st = check(c, A, B)
                assert st["rows_hash"] > 0
            finally:
                c.close()


@pytest.mark.parametrize("rpw_sym,rpw_num", [(3, 3), (16, 8), (5, 64)])
def test_short_rows_several_per_wavefront(rpw_sym, rpw_num):
    """The short-row kernels with several rows per wavefront (a small input takes one per wavefront
    unless rpw_fill is 0): the symbolic one takes consecutive rows, the numeric one rows a launch-wide
    stride apart, both with the next rows' loads in flight; row counts that leave the last
    wavefronts with fewer rows than the others."""
    for n, per_row in ((4099, 12), (1000, 16), (37, 5)):
        A = hostcsr.erdos_renyi(n, per_row, seed=n)
        c = capi.Context(0)
        try:
            c.set_option("window_max_span", 0)  # nothing takes the window path
            c.set_option("rpw_fill", 0)
            c.set_option("short_rpw", rpw_sym)
            c.set_option("short_num_rpw", rpw_num)
            c.set_option("profile_launches", 1)
            _, st = check(c, A, A)
            assert st["rows_hash"] > 0
            assert any(l["phase"] == "numeric" and l["path"] == "hash" and l["lanes"] == 1 for l in c.launches())
        finally:
            c.close()


def test_rmat_default_paths(ctx):
    A = hostcsr.rmat(13, 8, seed=1)
    check(ctx, A, A)


def test_rmat_scale17_natural_big_rows(ctx):
    """R-MAT scale 17, edge factor 8, default options: 13,741 rows of C have more than 4,096 entries
    (the longest 54,278); with 131,072 columns the rows leave the window path and the big ones go
    through the big-row kernels (bitmap windows; lists, slices and hub rows in the numeric pass)."""
    A = hostcsr.rmat(17, 8, seed=1)
    C, st = check(ctx, A, A)
    assert st["rows_global"] > 8000 and C.nnz == 183630319  # 9,197 rows with the hash taking up to 8,192 products


def test_rmat_forced_hash_and_global(ctx):
    A = hostcsr.rmat(12, 8, seed=2)
    ctx.set_option("window_max_span", 0)
    try:
        _, st = check(ctx, A, A)
        assert st["rows_hash"] > 0 and st["rows_global"] > 0 and st["rows_window"] == 0
    finally:
        ctx.set_option("window_max_span", 65536)


def test_rmat_workgroup_hash_tier_of_8192_products(ctx):
    """Rows of 4,097..8,192 products whose B rows are long take the largest hash tier: one table of
    128 KiB per workgroup of 1,024 threads."""
    A = hostcsr.rmat(15, 8, seed=4)
    ctx.set_option("window_max_span", 0)
    ctx.set_option("hash_max_nprod", 8192)
    ctx.set_option("profile_launches", 1)
    try:
        _, st = check(ctx, A, A)
        assert any(l["phase"] == "symbolic" and l["path"] == "hash" and l["lanes"] == 1024 for l in ctx.launches())
        assert st["rows_global"] > 0
    finally:
        ctx.set_option("window_max_span", 65536)
        ctx.set_option("hash_max_nprod", HASH_MAX_NPROD_DEFAULT)
        ctx.set_option("profile_launches", 0)


def test_rmat_forced_global_only(ctx):
    A = hostcsr.rmat(10, 8, seed=3)
    ctx.set_option("window_max_span", 0)
    ctx.set_option("hash_max_nprod", 0)
    try:
        _, st = check(ctx, A, A)
        assert st["rows_global"] > 0 and st["rows_hash"] == 0
    finally:
        ctx.set_option("window_max_span", 65536)
        ctx.set_option("hash_max_nprod", HASH_MAX_NPROD_DEFAULT)


def test_shell_like_small(ctx):
    A = hostcsr.shell_like(n_nodes=3000, grid_w=50)
    _, st = check(ctx, A, A)
    assert st["rows_window"] in (A.rows, A.rows // 5)  # symbolic runs once per group of identical rows


def test_diel_like_small(ctx):
    """3-D grid stand-in of dielFilterV3real: 27-point neighbourhoods, 3 unknowns per node."""
    A = hostcsr.diel_like(9, 8, 7)
    _, st = check(ctx, A, A)
    assert st["rows_window"] > 0


def test_exact_cancellation_is_dropped(ctx):
    # row 0 of A*B: column 0 gets 1*1 + 1*(-1) = 0 exactly -> erased; column 1 survives.
    A = hostcsr.HostCsr.from_coo(3, 2, [0, 0, 1, 2], [0, 1, 0, 1], [1.0, 1.0, 2.0, 3.0])
    B = hostcsr.HostCsr.from_coo(2, 3, [0, 0, 1, 1, 1], [0, 1, 0, 1, 2], [1.0, 1.0, -1.0, 5.0, 0.25])
    C, st = check(ctx, A, B)
    assert st["nnz_structural"] == C.nnz + 1
    # a larger case: B = [M ; -M] and A = [I I] cancel everything, plus a surviving diagonal
    n = 700
    M = hostcsr.banded_random(n, 5, 20, seed=11)
    r = M.row_ids()
    Br = np.concatenate([r, r + n, np.arange(n) + 2 * n])
    Bc = np.concatenate([M.col_idx, M.col_idx, np.arange(n)]).astype(np.int64)
    Bv = np.concatenate([M.val, -M.val, np.full(n, 2.0)])
    Bm = hostcsr.HostCsr.from_coo(3 * n, n, Br, Bc, Bv)
    Ar = np.tile(np.arange(n), 3)
    Ac = np.concatenate([np.arange(n), np.arange(n) + n, np.arange(n) + 2 * n])
    Am = hostcsr.HostCsr.from_coo(n, 3 * n, Ar, Ac, np.ones(3 * n))
    C, st = check(ctx, Am, Bm)
    assert C.nnz == n and st["nnz_structural"] > n


def test_empty_and_ragged(ctx):
    # no entries at all
    Z = hostcsr.HostCsr(5, 5, np.zeros(6, dtype=np.uint64), np.zeros(0, dtype=np.uint32), np.zeros(0))
    C, st = check(ctx, Z, Z)
    assert C.nnz == 0
    # empty rows in A, empty rows in B, a dense row, a single column
    n = 300
    r = np.concatenate([np.full(n, 7), np.arange(0, n, 3)])
    c = np.concatenate([np.arange(n), np.full(len(np.arange(0, n, 3)), 5)])
    v = np.arange(1, len(r) + 1, dtype=np.float64)
    A = hostcsr.HostCsr.from_coo(n, n, r, c, v)
    check(ctx, A, A)
    # rectangular
    A = hostcsr.HostCsr.from_coo(2, 3, [1, 0, 1], [0, 1, 2], [4.0, 2.0, 6.0])
    B = hostcsr.HostCsr.from_coo(3, 1, [0, 2], [0, 0], [1.0, -1.0])
    check(ctx, A, B)


def test_unsorted_input_rows_keep_cpu_order(ctx):
    # rows stored in descending column order: the k order differs from the sorted one and the
    # result must still follow the storage order bit for bit
    n = 2000
    M = hostcsr.banded_random(n, 12, 60, seed=5)
    rid = M.row_ids()
    order = np.lexsort((-M.col_idx.astype(np.int64), rid))
    A = hostcsr.HostCsr(n, n, M.row_ptr, M.col_idx[order], M.val[order])
    check(ctx, A, A)


def test_invalid_inputs_are_rejected(ctx):
    with pytest.raises(capi.SpgemmError):
        ctx.from_host(2, 2, np.array([0, 1, 2], dtype=np.uint64), np.array([0, 5], dtype=np.uint32), np.ones(2))
    A = hostcsr.HostCsr.from_coo(2, 3, [0], [0], [1.0])
    dA = ctx.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
    with pytest.raises(capi.SpgemmError):
        ctx.multiply(dA, dA)  # 2x3 times 2x3
    dA.free()


def test_section_bitmap_matches_oracle(ctx, golden_dir):
    A = hostcsr.read_mtx(f"{golden_dir}/ref_first/A.mtx")
    for P in (2, 4):
        parts = capi.partition_naive(A.rows, A.cols, P)
        for (r0, r1, _, _) in parts:
            blk = to_ocsr(A).select_rows(np.arange(r0, r1))
            d = ctx.from_host(blk.rows, blk.cols, blk.row_ptr, blk.col_idx, blk.val)
            assert np.array_equal(ctx.section_bitmap(d, 4), orc.section_bitmap(blk, 4))
            d.free()
    big = hostcsr.erdos_renyi(5000, 4, seed=9)
    blk = to_ocsr(big).select_rows(np.arange(100, 160))
    d = ctx.from_host(blk.rows, blk.cols, blk.row_ptr, blk.col_idx, blk.val)
    assert np.array_equal(ctx.section_bitmap(d, 512), orc.section_bitmap(blk, 512))
    with pytest.raises(capi.SpgemmError):
        ctx.section_bitmap(d, 4096 * 2)
    d.free()


@pytest.mark.parametrize("dofs", [2, 5, 8, 13, 20, 30, 70])
def test_row_groups_identical_patterns(ctx, dofs):
    # every node has `dofs` unknowns whose rows share one column list: runs of identical rows of
    # length dofs (shorter, equal and longer than a group, and longer than the run limit)
    nodes = max(40, 600 // dofs)
    P = hostcsr.banded_random(nodes, 4, 6, seed=dofs)
    rid, cid = P.row_ids(), P.col_idx.astype(np.int64)
    r = (rid[:, None, None] * dofs + np.arange(dofs)[None, :, None]).repeat(dofs, axis=2)
    c = (cid[:, None, None] * dofs + np.arange(dofs)[None, None, :]).repeat(dofs, axis=1)
    rng = np.random.default_rng(dofs)
    v = rng.random(r.size) * 2 - 1
    A = hostcsr.HostCsr.from_coo(nodes * dofs, nodes * dofs, r.ravel(), c.ravel(), v)
    _, st = check(ctx, A, A)
    ctx.set_option("row_groups", 0)
    try:
        check(ctx, A, A)
    finally:
        ctx.set_option("row_groups", 1)
    # one row / one group per wavefront instead of several with the next one's description in flight
    ctx.set_option("sym_rpw", 1)
    ctx.set_option("grp_rpw", 1)
    try:
        check(ctx, A, A)
    finally:
        ctx.set_option("sym_rpw", 8)
        ctx.set_option("grp_rpw", 8)
    # several per wavefront although the input is small (what a full-size matrix gets): 8, and 3 (a last
    # wavefront with fewer)
    ctx.set_option("rpw_fill", 0)
    try:
        check(ctx, A, A)
        ctx.set_option("sym_rpw", 3)
        ctx.set_option("grp_rpw", 3)
        check(ctx, A, A)
    finally:
        ctx.set_option("sym_rpw", 8)
        ctx.set_option("grp_rpw", 8)
        ctx.set_option("rpw_fill", 16)


def _node_blocks(P, dofs, rng):
    """Every entry of the node matrix P becomes a dofs x dofs block: runs of `dofs` rows with one column list."""
    rid, cid = P.row_ids(), P.col_idx.astype(np.int64)
    r = (rid[:, None, None] * dofs + np.arange(dofs)[None, :, None]).repeat(dofs, axis=2)
    c = (cid[:, None, None] * dofs + np.arange(dofs)[None, None, :]).repeat(dofs, axis=1)
    v = rng.random(r.size) * 2 - 1
    return hostcsr.HostCsr.from_coo(P.rows * dofs, P.cols * dofs, r.ravel(), c.ravel(), v)


@pytest.mark.parametrize("dofs,a_nodes,b_nodes", [(3, 40, 70), (2, 60, 100), (5, 24, 60)])
def test_row_runs_more_than_64_units_per_half(ctx, dofs, a_nodes, b_nodes):
    # A rows of up to 128 entries that select runs of long B rows (several chunks of 64 columns each):
    # more than 64 units in the first and in the second half of the A row, so the unit tables of the run
    # walk are refilled, in the symbolic pass and in the numeric pass of the wide groups
    rng = np.random.default_rng(100 * dofs + a_nodes)
    nodes = min(140, 500 // dofs)  # (at most 512 outputs per row: the rows stay grouped)

    def node_matrix(per_row):
        r = np.repeat(np.arange(nodes), per_row)
        c = np.concatenate([np.sort(rng.choice(nodes, per_row, replace=False)) for _ in range(nodes)])
        return hostcsr.HostCsr.from_coo(nodes, nodes, r, c, np.ones(len(r)))

    A = _node_blocks(node_matrix(a_nodes), dofs, rng)
    B = _node_blocks(node_matrix(b_nodes), dofs, rng)
    assert 64 < A.nnz // A.rows <= 128 and B.nnz // B.rows > 128
    for fill in (16, 0):
        ctx.set_option("rpw_fill", fill)
        ctx.set_option("profile_launches", 1)
        try:
            check(ctx, A, B)
            # (the case must reach the kernels it is written for)
            assert any(l["phase"] == "numeric" and l["path"] == "window_group_runs" for l in ctx.launches())
        finally:
            ctx.set_option("rpw_fill", 16)
            ctx.set_option("profile_launches", 0)


def test_row_runs_half_boundary_and_empty_b_rows(ctx):
    # A rows of 1 .. 128 entries that select CONSECUTIVE rows of a B made of blocks of identical rows
    # (some blocks empty): runs that cross the 64th entry, runs cut by the row's end, empty B rows
    # inside a run, rows of exactly 64 / 65 / 128 entries; every A row twice, so that the group
    # kernels take them too
    rng = np.random.default_rng(77)
    nb, cols, dof = 480, 300, 5
    br, bc = [], []
    for blk in range(nb // dof):
        L = 0 if blk % 7 == 3 else int(rng.integers(3, 91))
        cl = np.sort(rng.choice(cols, L, replace=False))
        for i in range(dof):
            br.append(np.full(L, blk * dof + i))
            bc.append(cl)
    br, bc = np.concatenate(br), np.concatenate(bc)
    B = hostcsr.HostCsr.from_coo(nb, cols, br, bc, rng.random(len(br)) * 2 - 1)
    ar, ac = [], []
    row = 0
    for ln in (1, 2, 5, 6, 59, 63, 64, 65, 66, 69, 100, 127, 128):
        for start in (0, 3, int(rng.integers(0, nb - 128))):
            ks = np.arange(start, start + ln)
            for _ in range(2):
                ar.append(np.full(ln, row))
                ac.append(ks)
                row += 1
    ar, ac = np.concatenate(ar), np.concatenate(ac)
    A = hostcsr.HostCsr.from_coo(row, nb, ar, ac, rng.random(len(ar)) * 2 - 1)
    for fill in (16, 0):
        ctx.set_option("rpw_fill", fill)
        try:
            check(ctx, A, B)
        finally:
            ctx.set_option("rpw_fill", 16)


def test_row_runs_units_of_128_columns_boundaries(ctx):
    # B made of blocks of identical rows whose lengths sit on either side of the 64- and 128-column
    # marks (63 .. 257: one to three units of 128 columns, the second half of a unit empty, partly or
    # wholly filled), with more than 64 entries per row on average, so that the symbolic run kernel
    # takes units of 128 columns as the numeric kernel of the wide groups always does; A rows of 1 .. 128
    # entries that select consecutive B rows, every A row twice (groups)
    rng = np.random.default_rng(128)
    dof, cols = 3, 420
    lens = [63, 64, 65, 127, 128, 129, 191, 192, 193, 256, 257, 1, 130, 66, 200, 100] * 3
    nb = len(lens) * dof
    br, bc = [], []
    for blk, L in enumerate(lens):
        cl = np.sort(rng.choice(cols, L, replace=False))
        for i in range(dof):
            br.append(np.full(L, blk * dof + i))
            bc.append(cl)
    br, bc = np.concatenate(br), np.concatenate(bc)
    B = hostcsr.HostCsr.from_coo(nb, cols, br, bc, rng.random(len(br)) * 2 - 1)
    assert B.nnz > 64 * B.rows
    ar, ac = [], []
    row = 0
    for ln in (1, 3, 6, 33, 63, 64, 65, 66, 96, 127, 128):
        for start in (0, 2, int(rng.integers(0, nb - 128))):
            ks = np.arange(start, start + ln)
            for _ in range(2):
                ar.append(np.full(ln, row))
                ac.append(ks)
                row += 1
    ar, ac = np.concatenate(ar), np.concatenate(ac)
    A = hostcsr.HostCsr.from_coo(row, nb, ar, ac, rng.random(len(ar)) * 2 - 1)
    for fill in (16, 0):
        ctx.set_option("rpw_fill", fill)
        ctx.set_option("profile_launches", 1)
        try:
            check(ctx, A, B)
            assert any(l["phase"] == "numeric" and l["path"] == "window_group_runs" for l in ctx.launches())
        finally:
            ctx.set_option("rpw_fill", 16)
            ctx.set_option("profile_launches", 0)


def test_row_groups_with_cancellation_and_order(ctx):
    # identical column lists but stored in a different order must NOT be grouped (the k order
    # differs); identical rows whose values cancel exactly are dropped per row
    n = 64
    r = np.repeat(np.arange(n), 3)
    c = np.tile(np.array([1, 5, 9]), n)
    c[3:6] = [9, 5, 1]  # row 1: same set, other order
    v = np.ones(3 * n)
    v[6:9] = [1.0, -1.0, 0.5]  # row 2
    A = hostcsr.HostCsr.from_coo(n, n, r, c, v)
    Br = np.array([1, 5, 9, 9])
    Bc = np.array([0, 0, 0, 3])
    Bv = np.array([2.0, 2.0, 7.0, 1.0])
    B = hostcsr.HostCsr.from_coo(n, n, Br, Bc, Bv)
    C, st = check(ctx, A, B)
    assert st["nnz_structural"] >= C.nnz


def _big_row_matrix(seed):
    """R-MAT plus a few dense-ish hub rows: output rows of several thousand entries, A rows longer
    than one wavefront and than the in-LDS cursor table."""
    A = hostcsr.rmat(13, 8, seed=seed)
    rng = np.random.default_rng(seed)
    n = A.rows
    r, c, v = [A.row_ids()], [A.col_idx.astype(np.int64)], [A.val]
    for hub, ln in ((5, 3000), (77, 1500), (n - 3, 200), (n // 2, 70)):
        cols = rng.choice(n, size=ln, replace=False)
        r.append(np.full(ln, hub)); c.append(cols); v.append(rng.uniform(-1, 1, ln))
    return hostcsr.HostCsr.from_coo(n, n, np.concatenate(r), np.concatenate(c), np.concatenate(v),
                                   sum_duplicates=True).sorted_rows()


@pytest.mark.parametrize("win_words", [32768, 1024])
def test_big_rows_slices_and_windows(ctx, win_words):
    """Big-row kernels: several bitmap windows per row (small window forced), output rows cut into
    many slices, long A rows (list in global scratch, runs of several chunks per item)."""
    A = _big_row_matrix(9)
    ctx.set_option("window_max_span", 0)
    ctx.set_option("hash_max_nprod", 64)
    ctx.set_option("big_win_words", win_words)
    try:
        _, st = check(ctx, A, A)
        assert st["rows_global"] > 1000
    finally:
        ctx.set_option("window_max_span", 65536)
        ctx.set_option("hash_max_nprod", HASH_MAX_NPROD_DEFAULT)
        ctx.set_option("big_win_words", 32768)


@pytest.mark.parametrize("order", [1, 2, 3])
def test_slice_items_sorted_many_rows_of_one_chunk(order):
    """Thousands of rows of about a hundred outputs in the big-row numeric kernel (hash_num_slice = 64), one
    item each, the items sorted before the kernel takes them: more items than a quarter of the chunks plus
    two per row would hold when rows have FEWER than four chunks - the count the sort once ran over.  A fresh
    context per run: the first multiply writes into new memory, where a lost item shows."""
    A = hostcsr.erdos_renyi(3000, 10, seed=31)
    for _ in range(3):
        c = capi.Context(0)
        try:
            c.set_option("window_max_span", 0)
            c.set_option("short_max_nprod", 0)
            c.set_option("hash_num_slice", 64)
            c.set_option("slice_sorted", order)
            c.set_option("profile_launches", 1)
            check(c, A, A)
            assert any(l["phase"] == "numeric" and l["path"] == "big_rows" and l["rows"] > 2000 for l in c.launches())
        finally:
            c.close()


def test_big_rows_few_workgroups(ctx):
    """The slice kernel hands out items through a counter: one workgroup must drain them all."""
    A = _big_row_matrix(10)
    ctx.set_option("window_max_span", 0)
    ctx.set_option("slice_blocks", 1)
    try:
        check(ctx, A, A)
        ctx.set_option("slice_blocks", 3)
        check(ctx, A, A)
    finally:
        ctx.set_option("window_max_span", 65536)
        ctx.set_option("slice_blocks", 0)


def test_unsorted_b_rows_take_the_general_path(ctx):
    """B rows that are not sorted by column: the big-row kernels (which rely on sorted B rows) are
    not used, the result is the same."""
    A = hostcsr.rmat(11, 8, seed=5)
    rng = np.random.default_rng(0)
    col, val = A.col_idx.copy(), A.val.copy()
    rp = A.row_ptr.astype(np.int64)
    for r in range(A.rows):  # random order inside every row
        p = rng.permutation(rp[r + 1] - rp[r]) + rp[r]
        col[rp[r]:rp[r + 1]] = A.col_idx[p]
        val[rp[r]:rp[r + 1]] = A.val[p]
    Bu = hostcsr.HostCsr(A.rows, A.cols, A.row_ptr, col, val)
    ctx.set_option("window_max_span", 0)
    try:
        check(ctx, A, Bu)
    finally:
        ctx.set_option("window_max_span", 65536)


def test_explicit_zero_entries_follow_the_value_rule(ctx):
    """Stored 0.0 entries in A and B.  The library keeps an output entry iff its accumulated value
    is != 0 (the rule of scipy, which wrote every C_expected.mtx).  The reference's release build
    differs in one corner: a product that is exactly 0.0 and arrives as the FIRST product of an
    entry is inserted and stays (include/matrix_impl.hpp:83-88, the assert is compiled out), so the
    oracle's result holds explicit zeros there.  Apart from those zero-valued entries the two
    results are the same, bit for bit."""
    rng = np.random.default_rng(17)
    n = 400
    M = hostcsr.banded_random(n, 7, 25, seed=23)
    va = M.val.copy()
    va[rng.random(len(va)) < 0.3] = 0.0          # explicit zeros in A
    A = hostcsr.HostCsr(n, n, M.row_ptr, M.col_idx, va)
    vb = M.val.copy()
    vb[rng.random(len(vb)) < 0.3] = 0.0          # explicit zeros in B
    B = hostcsr.HostCsr(n, n, M.row_ptr, M.col_idx, vb)
    C, st = gpu_multiply(ctx, A, B)
    E = orc.spgemm(to_ocsr(A), to_ocsr(B))
    assert np.all(C.val != 0.0)
    zero = E.val == 0.0
    assert zero.any(), "the case must exercise the corner"
    keep = ~zero
    rows_of = np.repeat(np.arange(E.rows), np.diff(E.row_ptr.astype(np.int64)))
    rp = np.zeros(E.rows + 1, dtype=np.uint64)
    rp[1:] = np.cumsum(np.bincount(rows_of[keep], minlength=E.rows))
    orc.assert_same(C, orc.OCsr(E.rows, E.cols, rp, E.col_idx[keep], E.val[keep]), rtol=0.0)
    assert st["nnz_c"] == int(keep.sum())


def test_wrapped_b_refilled_in_place_needs_invalidate(ctx):
    """A caller that wraps its receive buffers once and refills them every run (as the reference
    reuses its buffers) calls spgemm_mat_invalidate: the cached per-row facts of B (column range,
    length, sortedness) are rebuilt."""
    import torch
    n = 3000
    M1 = hostcsr.banded_random(n, 9, 40, seed=31)
    # same row lengths, mirrored columns stored in DESCENDING order: other column ranges, rows unsorted
    M2 = hostcsr.HostCsr(n, n, M1.row_ptr, (n - 1 - M1.col_idx.astype(np.int64)).astype(np.uint32), M1.val * 0.5)
    t_ptr = torch.from_numpy(M1.row_ptr.astype(np.int64)).cuda()
    t_col = torch.from_numpy(M1.col_idx.view(np.int32).copy()).cuda()
    t_val = torch.from_numpy(M1.val.copy()).cuda()
    torch.cuda.synchronize()
    dB = ctx.wrap_device(n, n, M1.nnz, t_ptr.data_ptr(), t_col.data_ptr(), t_val.data_ptr(), validate=True,
                         keepalive=(t_ptr, t_col, t_val))
    dA = ctx.from_host(M1.rows, M1.cols, M1.row_ptr, M1.col_idx, M1.val)

    def run():
        dC = ctx.multiply(dA, dB)
        rp, ci, v = dC.to_host()
        dC.free()
        return orc.OCsr(n, n, rp, ci, v)

    orc.assert_same(run(), orc.spgemm(to_ocsr(M1), to_ocsr(M1)), rtol=0.0)
    t_col.copy_(torch.from_numpy(M2.col_idx.view(np.int32).copy()))
    t_val.copy_(torch.from_numpy(M2.val.copy()))
    torch.cuda.synchronize()
    dB.invalidate()
    orc.assert_same(run(), orc.spgemm(to_ocsr(M1), to_ocsr(M2)), rtol=0.0)
    dB.free()
    dA.free()


def test_lane_order_probe_and_both_add_modes():
    """The device passes the probe of spgemm_create (one LDS fp64 add applies lanes that share an
    address in lane order), and the flat-walk kernels give the oracle's bits with the
    single-instruction adds and with claim rounds alike.  The matrix makes many products of one
    step hit the same output: every row of A refers to rows of B that all hold the same few
    columns, with values of very different magnitude, so that any other order of the additions
    rounds differently."""
    rng = np.random.default_rng(77)
    n, k = 600, 400
    r = np.repeat(np.arange(n), 40)
    c = rng.integers(0, k, len(r))
    v = np.ldexp(1.0 + rng.random(len(r)), rng.integers(-30, 31, len(r))) * rng.choice([-1.0, 1.0], len(r))
    A = hostcsr.HostCsr.from_coo(n, k, r, c, v, sum_duplicates=True).sorted_rows()
    hot = np.array([3, 17, 18, 250, 251, 252, 399])  # the same few columns in every row of B
    rb = np.repeat(np.arange(k), len(hot))
    cb = np.tile(hot, k)
    vb = np.ldexp(1.0 + rng.random(len(rb)), rng.integers(-30, 31, len(rb)))
    B = hostcsr.HostCsr.from_coo(k, k, rb, cb, vb, sum_duplicates=True).sorted_rows()
    E = orc.spgemm(to_ocsr(A), to_ocsr(B))
    seen = []
    for mode in (1, 0):
        for hash_path in (0, 1):
            c2 = capi.Context(0)
            try:
                assert c2.device_property("cu_count") > 0
                seen.append(c2.device_property("lane_order_probe"))
                c2.set_option("lane_order_adds", mode)
                c2.set_option("row_groups", 0)
                c2.set_option("short_max_nprod", 0)
                if hash_path:
                    c2.set_option("window_max_span", 0)
                dA = c2.from_host(A.rows, A.cols, A.row_ptr, A.col_idx, A.val)
                dB = c2.from_host(B.rows, B.cols, B.row_ptr, B.col_idx, B.val)
                dC = c2.multiply(dA, dB)
                rp, ci, vv = dC.to_host()
                orc.assert_same(orc.OCsr(A.rows, B.cols, rp, ci, vv), E, rtol=0.0)
            finally:
                c2.close()
    if not all(s == 1 for s in seen):  # not an error: the kernels then take the claim rounds, checked above
        import warnings
        warnings.warn("this device does not apply the lanes of one LDS fp64 add in lane order")
