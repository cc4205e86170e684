"""The reference's own doctests / unit tests for crates/vdb, run against BOTH the oracle restatement
(oracle/vdb.c) and the product tree builder (dust_amd/csrc/vdb.cpp through the C ABI).
These are the only known-answer tests the reference holds for this path."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from dust_amd import _lib as L
from dust_amd import api


class OracleTree:
    def __init__(self, *log2s):
        self.l = O.lib()
        self.h = self.l.orc_tree_new((C.c_uint32 * len(log2s))(*log2s), len(log2s))

    def __del__(self):
        self.l.orc_tree_free(self.h)

    def set_value(self, p, v):
        self.l.orc_tree_set(self.h, p[0], p[1], p[2], -1 if v is None else int(v))

    def get_value(self, p):
        r = self.l.orc_tree_get(self.h, p[0], p[1], p[2])
        return None if r < 0 else bool(r)

    def iter(self):
        n = self.l.orc_tree_iter(self.h, None, 0)
        out = np.zeros((max(n, 1), 3), np.uint32)
        self.l.orc_tree_iter(self.h, out.ctypes.data_as(C.c_void_p), n)
        return out[:n]

    def iter_leaf(self):
        n = self.l.orc_tree_iter_leaf(self.h, None, None, None, 0)
        xyz, occ, mp = np.zeros((max(n, 1), 3), np.uint32), np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint32)
        self.l.orc_tree_iter_leaf(self.h, xyz.ctypes.data_as(C.c_void_p), occ.ctypes.data_as(C.c_void_p), mp.ctypes.data_as(C.c_void_p), n)
        return xyz[:n], occ[:n], mp[:n]

    def meta(self):
        return self.l.orc_tree_meta_mask(self.h), self.l.orc_tree_root_level(self.h)

    def accessor(self):
        tree = self

        class A:
            def __init__(s):
                s.h = tree.l.orc_accessor_new(tree.h)

            def get(s, p):
                r = tree.l.orc_accessor_get(s.h, p[0], p[1], p[2])
                return None if r < 0 else bool(r)

            def __del__(s):
                tree.l.orc_accessor_free(s.h)
        return A()


IMPLS = [pytest.param(OracleTree, id="oracle"), pytest.param(api.Tree, id="product")]


@pytest.mark.parametrize("Tree", IMPLS)
def test_tree_get_set_doctest(Tree):
    # crates/vdb/src/tree.rs:15-25
    t = Tree(2, 2)
    t.set_value((0, 4, 0), True)
    t.set_value((0, 2, 2), False)
    assert t.get_value((0, 4, 0)) is True
    assert t.get_value((0, 3, 0)) is None
    assert t.get_value((0, 2, 2)) is False


@pytest.mark.parametrize("Tree", IMPLS)
def test_tree_iter_order_doctest(Tree):
    # crates/vdb/src/tree.rs:87-101
    t = Tree(4, 2)
    for p in ((0, 1, 2), (63, 1, 3), (63, 63, 63)):
        t.set_value(p, True)
    assert t.iter().tolist() == [[0, 1, 2], [63, 1, 3], [63, 63, 63]]


def test_bitmask_doctest():
    # crates/vdb/src/bitmask.rs:81-90
    for setter, it in ((O.lib().orc_bitmask_set, O.lib().orc_bitmask_iter),
                       (L.load().dust_vdb_bitmask_set, L.load().dust_vdb_bitmask_iter_set_bits)):
        words = np.zeros(2, np.uint64)
        p = words.ctypes.data_as(C.POINTER(C.c_uint64)) if setter is not O.lib().orc_bitmask_set else words.ctypes.data_as(C.c_void_p)
        setter(p, 12, 1)
        setter(p, 101, 1)
        out = np.zeros(8, np.uint32)
        po = out.ctypes.data_as(C.POINTER(C.c_uint32)) if setter is not O.lib().orc_bitmask_set else out.ctypes.data_as(C.c_void_p)
        n = it(p, 2, po, 8)
        assert n == 2 and out[:2].tolist() == [12, 101]


def test_pool_doctest():
    # crates/vdb/src/pool.rs:22-42
    o = O.lib()
    p = o.orc_pool_new(8, 1)
    assert [o.orc_pool_alloc(p) for _ in range(4)] == [0, 1, 2, 3]
    assert o.orc_pool_num_chunks(p) == 2
    o.orc_pool_free(p, 1)
    o.orc_pool_free(p, 2)
    assert [o.orc_pool_alloc(p) for _ in range(3)] == [2, 1, 4]
    o.orc_pool_free_pool(p)
    l = L.load()
    h = C.c_void_p()
    L.check(l.dust_vdb_pool_create(8, 1, C.byref(h)))
    assert [l.dust_vdb_pool_alloc(h) for _ in range(4)] == [0, 1, 2, 3]
    assert l.dust_vdb_pool_num_chunks(h) == 2
    l.dust_vdb_pool_free(h, 1)
    l.dust_vdb_pool_free(h, 2)
    assert [l.dust_vdb_pool_alloc(h) for _ in range(3)] == [2, 1, 4]
    l.dust_vdb_pool_destroy(h)


@pytest.mark.parametrize("log2s", [(2,), (3, 1), (2, 2, 1)])
def test_hierarchy_forms(log2s):
    # crates/vdb/src/node/mod.rs:100-111 (the forms with a fixed root)
    assert OracleTree(*log2s).h
    assert api.Tree(*log2s)._h


@pytest.mark.parametrize("Tree", IMPLS)
def test_meta_mask_and_lca(Tree):
    # crates/vdb/src/accessor.rs:148-170
    t = Tree(2, 4, 2)
    mask, root_level = t.meta()
    assert mask == 0b10100010
    assert root_level == 2
    if Tree is OracleTree:
        a = (C.c_uint32 * 3)(0, 0, 0)
        b = (C.c_uint32 * 3)(255, 255, 255)
        assert O.lib().orc_lca_level(a, b, mask, root_level) == 2
    else:
        assert api.lca_level((0, 0, 0), (255, 255, 255), mask, root_level) == 2


@pytest.mark.parametrize("Tree", IMPLS)
def test_accessor_random(Tree):
    # crates/vdb/src/accessor.rs:172-195 (seeded instead of thread_rng)
    rng = np.random.default_rng(20241008)
    t = Tree(2, 4, 2)
    locs = rng.integers(0, 256, size=(100, 3))
    for p in locs:
        t.set_value(tuple(int(v) for v in p), True)
    acc = t.accessor()
    for i in rng.permutation(100):
        assert acc.get(tuple(int(v) for v in locs[i])) is True


def test_product_matches_oracle_on_random_trees():
    rng = np.random.default_rng(5)
    for log2s in ((4, 2, 2), (2, 2), (3, 2, 2), (4, 4, 2, 2)):
        ext = 1 << sum(log2s)
        a, b = OracleTree(*log2s), api.Tree(*log2s)
        pts = rng.integers(0, min(ext, 512), size=(3000, 3))
        vals = rng.integers(0, 2, 3000)
        for p, v in zip(pts, vals):
            p = tuple(int(x) for x in p)
            a.set_value(p, bool(v))
            b.set_value(p, bool(v))
        assert np.array_equal(a.iter(), b.iter())
        xa, oa, _ = a.iter_leaf()
        xb, ob, _ = b.iter_leaf()
        assert np.array_equal(xa, xb) and np.array_equal(oa, ob)
        q = rng.integers(0, min(ext, 512), size=(500, 3))
        acc_a, acc_b = a.accessor(), b.accessor()
        for p in q:
            p = tuple(int(x) for x in p)
            assert a.get_value(p) == b.get_value(p)
        for p in np.concatenate([pts[:200], q[:50]]):
            p = tuple(int(x) for x in p)
            if a.get_value(p) is not None:  # the accessor's cached path is only defined after successful gets
                assert acc_a.get(p) == acc_b.get(p) == a.get_value(p)


def test_clear_is_unsupported_like_reference():
    # crates/vdb/src/node/internal.rs:121-124 is todo!()
    t = api.Tree(4, 2, 2)
    with pytest.raises(L.DustError) as e:
        t.set_value((1, 2, 3), None)
    assert e.value.status == L.ERR_UNSUPPORTED
