"""GPU (-m gpu): sort and scan primitives beyond the reference's vectors -- edge sizes, stability,
skewed digits, full-size properties.  Bit-exact against numpy (stable argsort / cumsum) and the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gpe):
    c = gpe.Context()
    yield c
    c.close()


def _sort(gpe, ctx, keys, vals):
    kb, vb = gpe.GpuBuffer(ctx, keys), gpe.GpuBuffer(ctx, vals)
    ctx.call("gpe_sort_pairs_u32", kb.dptr, vb.dptr, len(keys))
    k, v = kb.download().copy(), vb.download().copy()
    kb.free(); vb.free()
    return k, v


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 1023, 4095, 4096, 4097, 8191, 8193, 25006, 1 << 20])
def test_sort_sizes_random_keys_stable(gpe, ctx, n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    k, v = _sort(gpe, ctx, keys, vals)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order])
    assert np.array_equal(v, vals[order])


def test_sort_across_the_tile_size_switch(gpe, ctx):
    """Sorts of up to 3 * 2^20 keys run on 4096-key tiles, larger ones on 8192-key tiles (k_onesweep.hip); both
    share the epoch-tagged status array.  Sizes on either side of the switch, in an order that alternates the two."""
    for n in (3 << 20, (3 << 20) + 1, 100_003, (3 << 20) - 1, 5_000_011, 4097):
        rng = np.random.default_rng(n)
        keys = rng.integers(0, 1 << 22, n, dtype=np.uint64).astype(np.uint32)     # duplicates: stability matters
        vals = np.arange(n, dtype=np.uint32)
        k, v = _sort(gpe, ctx, keys, vals)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(k, keys[order]), n
        assert np.array_equal(v, vals[order]), n


@pytest.mark.parametrize("kind", ["all_equal", "two_values", "few_bits", "nearly_sorted", "unused_tail", "top_byte"])
def test_sort_skewed_digit_distributions(gpe, ctx, kind):
    n = 300_017
    rng = np.random.default_rng(5)
    if kind == "all_equal":
        keys = np.full(n, 0xDEADBEEF, np.uint32)
    elif kind == "two_values":
        keys = rng.choice(np.array([7, 0xFFFFFFFF], np.uint32), n)
    elif kind == "few_bits":
        keys = rng.integers(0, 4, n, dtype=np.uint32) << 9
    elif kind == "nearly_sorted":
        keys = (np.arange(n, dtype=np.uint32) // 3) + rng.integers(0, 5, n, dtype=np.uint32)
    elif kind == "unused_tail":
        keys = rng.integers(0, 1 << 20, n, dtype=np.uint32)
        keys[rng.random(n) < 0.3] = 0xFFFFFFFF
    else:
        keys = rng.integers(0, 256, n, dtype=np.uint32) << 24
    vals = np.arange(n, dtype=np.uint32)
    k, v = _sort(gpe, ctx, keys, vals)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order])
    assert np.array_equal(v, vals[order])          # stability: payload == original index order in ties


def test_sort_matches_oracle_pass_by_pass(gpe, ctx, oracle):
    n = 50_000
    rng = np.random.default_rng(11)
    keys = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    vals = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    for shift in (0, 8, 16, 24):
        ka, va = gpe.GpuBuffer(ctx, keys), gpe.GpuBuffer(ctx, vals)
        kb, vb = gpe.GpuBuffer(ctx, np.zeros(n, np.uint32)), gpe.GpuBuffer(ctx, np.zeros(n, np.uint32))
        hist = gpe.GpuBuffer(ctx, np.zeros(256, np.uint32))
        ctx.call("gpe_sort_histogram_u32", ka.dptr, n, shift, hist.dptr)
        ctx.call("gpe_sort_scatter_pass_u32", ka.dptr, va.dptr, kb.dptr, vb.dptr, n, shift)
        ohist = oracle.radix_build_histogram(keys, shift)
        ok, ov = oracle.radix_scatter(keys, vals, shift, ohist)
        assert np.array_equal(hist.download(), ohist.reshape(-1, 256).sum(0))
        assert np.array_equal(kb.download(), ok)
        assert np.array_equal(vb.download(), ov)
        for b in (ka, va, kb, vb, hist):
            b.free()


def test_sort_full_size_properties(gpe, ctx):
    """4M pairs (the 1M-particle grid sort): sortedness, permutation, stability -- size-independent checks."""
    n = 4_000_000
    rng = np.random.default_rng(3)
    keys = rng.integers(0, 1 << 22, n, dtype=np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    k, v = _sort(gpe, ctx, keys, vals)
    assert (np.diff(k.astype(np.int64)) >= 0).all()
    assert np.array_equal(keys[v], k)                         # payload still names its key
    assert np.array_equal(np.sort(v), vals)                   # a permutation
    ties = np.diff(k.astype(np.int64)) == 0
    assert (np.diff(v.astype(np.int64))[ties] > 0).all()      # stable


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 65535, 65536, 65537,
                               81920, 83090, (1 << 24) + 1, 4096 * 4096 + 5])
def test_scan_sizes(gpe, ctx, n):
    rng = np.random.default_rng(n % 1000)
    data = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)    # exercises wrap-around
    buf = gpe.GpuBuffer(ctx, data)
    ctx.call("gpe_inclusive_scan_u32", buf.dptr, n)
    assert np.array_equal(buf.download(), np.cumsum(data, dtype=np.uint64).astype(np.uint32))
    buf.free()


def test_scan_linearity_full_size(gpe, ctx):
    """scan(a) + scan(b) == scan(a + b) (mod 2^32) at the 100M-particle chunk-count size / 4."""
    n = 25_000_000
    rng = np.random.default_rng(8)
    a = rng.integers(0, 5, n, dtype=np.uint32)
    b = rng.integers(0, 5, n, dtype=np.uint32)
    out = []
    for x in (a, b, a + b):
        buf = gpe.GpuBuffer(ctx, x)
        ctx.call("gpe_inclusive_scan_u32", buf.dptr, n)
        out.append(buf.download().copy())
        buf.free()
    assert np.array_equal(out[0] + out[1], out[2])
    assert out[2][-1] == np.uint32((a.sum(dtype=np.uint64) + b.sum(dtype=np.uint64)) & 0xFFFFFFFF)


def test_gpu_buffer_push_replace_download_last(gpe, ctx):
    """GpuBuffer::push / push_all / replace_elem / download_last (utils/gpu_buffer.rs:30-38,177-275): the device
    buffer grows by doubling, keeps its DEVICE contents across the growth, and only the new tail is written."""
    buf = gpe.GpuBuffer(ctx, np.arange(5, dtype=np.uint32))
    assert buf.capacity_bytes() == 20 and buf.download_last() == 4
    # something on the device that the host mirror does not know about (a kernel wrote it): it survives the growth
    ctx.call("gpe_inclusive_scan_u32", buf.dptr, 5)                    # device: 0 1 3 6 10; mirror still 0 1 2 3 4
    assert buf.download_last() == 10 and list(buf.data()) == [0, 1, 2, 3, 4]
    buf.push(77)                                                       # 24 > 20 bytes: new buffer of 48
    assert buf.len() == 6 and buf.capacity_bytes() == 48
    assert buf.download_last() == 77
    buf.push_all([5, 6, 7])                                            # 36 <= 48: in place
    assert buf.capacity_bytes() == 48 and buf.len() == 9
    buf.replace_elem(1234, 2)
    assert list(buf.download()) == [0, 1, 1234, 6, 10, 77, 5, 6, 7]
    with pytest.raises(IndexError):
        buf.replace_elem(1, 9)
    big = np.arange(100000, dtype=np.uint32)
    buf.push_all(big)
    assert buf.capacity_bytes() == 2 * 4 * (9 + 100000)
    out = buf.download()
    assert list(out[:9]) == [0, 1, 1234, 6, 10, 77, 5, 6, 7] and np.array_equal(out[9:], big)
    assert buf.download_last() == 99999
    empty = gpe.GpuBuffer(ctx, np.zeros(0, np.uint32))
    assert empty.download_last() is None
    empty.push(3)
    assert empty.download_last() == 3 and list(empty.download()) == [3]
    empty.free()
    buf.free()
