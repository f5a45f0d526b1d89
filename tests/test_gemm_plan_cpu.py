"""Host logic of the GEMM dispatch, read from the library (utx_gemm_plan: pure arithmetic, no device): which kernel a FLUX linear goes to and how the
one-wave-per-SIMD kernel cuts a partly filled last round along K (gemm_w4.hip "split tail").  The range arithmetic of the kernel (W4_NEXT_SEG) and
of the fix-up kernel is restated here in a few lines and checked for what the GPU tests cannot show directly: every K-tile of every tail tile is
owned by exactly one range, ranges are never empty, partial-tile slots are unique and fit the workspace."""
import itertools

import pytest

from unitex_amd import _lib
from unitex_amd.flux import ops

NCU = 256


def _ranges(plan, nssu, grid=NCU):
    """(workgroup, pass) -> (tail tile, k0, k1, slot): W4_NEXT_SEG of gemm_w4.hip"""
    T, S = plan["tail_tiles"], plan["ranges"]
    out = []
    for w in range(grid):
        tj = 0
        while w + tj * grid < T * S:
            r = w + tj * grid
            j = r // T
            out.append((r - j * T, (j * nssu) // S, ((j + 1) * nssu) // S, r))
            tj += 1
    return out


@pytest.fixture(autouse=True)
def _default_options():
    yield
    for k, v in (("UTX_GEMM_STREAMK", 1), ("UTX_GEMM_TILE", 0), ("UTX_GEMM_PERS_GRID", 0)):
        _lib.set_option(k, v)


def test_plan_of_the_flux_linears():
    # the reference strip (13 376 computed rows), BASELINE's strip (50 240), the pruned last block (6144 rows), no text de-duplication (13 824)
    expect = {
        (13376, 3072, 12288, 64): ("w4", 636, 124, 2), (13376, 3072, 15360, 0): ("w4", 636, 124, 2), (13376, 3072, 3072, 64): ("w4", 636, 124, 2),
        (13376, 9216, 3072, 64): ("w4", 1908, 116, 2), (13376, 12288, 3072, 64): ("w4", 2544, 0, 0), (13376, 21504, 3072, 64): ("w4", 4452, 100, 2),
        (50240, 3072, 12288, 64): ("w4", 2364, 60, 4), (50240, 3072, 15360, 0): ("w4", 2364, 60, 4), (50240, 9216, 3072, 64): ("w4", 7092, 0, 0),
        (6144, 3072, 15360, 0): ("w4", 288, 32, 8), (13824, 3072, 12288, 64): ("w4", 648, 136, 3), (13824, 3072, 3072, 64): ("w4", 648, 0, 0),
        (64, 9216, 3072, 64): ("128x128", 72, 0, 0), (13376, 192, 3072, 0): ("128x128", 210, 0, 0),
    }
    for (M, N, K, K2), want in expect.items():
        p = ops.gemm_plan(M, N, K=K, K2=K2)
        assert (p["kernel"], p["tiles"], p["tail_tiles"], p["ranges"]) == want, ((M, N, K, K2), p)
    # no scratch, the option off, a fused q / k projection's non-uniform LoRA extent: the last round stays whole
    assert ops.gemm_plan(13376, 3072, K=12288, K2=64, sk=False)["tail_tiles"] == 0
    _lib.set_option("UTX_GEMM_STREAMK", 0)
    assert ops.gemm_plan(13376, 3072, K=12288, K2=64)["tail_tiles"] == 0
    _lib.set_option("UTX_GEMM_STREAMK", 1)
    assert ops.gemm_plan(13376, 21504, K=3072, K2=64, lora_n_limit=9216, lora_seg_n=3072, n_split=9216, gelu_from=9216)["tail_tiles"] == 0
    # kernel selection switches
    for tile, name in ((128, "128x128"), (2560, "pers8"), (256, "8phase"), (2562, "2barrier"), (2564, "w4")):
        _lib.set_option("UTX_GEMM_TILE", tile)
        assert ops.gemm_plan(13376, 3072, K=3072)["kernel"] == name
    _lib.set_option("UTX_GEMM_TILE", 0)
    assert ops.gemm_takes_w4(50240, 21504, n_split=9216, gelu_from=9216, K2=64, lora_seg_n=3072, lora_n_limit=9216)
    assert not ops.gemm_takes_w4(50240, 21504 + 128)          # a column boundary off the 256 grid


def test_split_ranges_tile_every_tail_tile_exactly_once():
    Ms = [256 * m + d for m in (24, 25, 37, 52, 53, 54, 96, 197, 198) for d in (0, -255, -3)]
    cases = 0
    for M, N, K, K2, force in itertools.product(Ms, (3072, 9216, 21504), (512, 3072, 12288, 15360), (0, 64, 512), (0, 2, 3, 5, 8)):
        _lib.set_option("UTX_GEMM_STREAMK", 1000 + force if force else 1)
        p = ops.gemm_plan(M, N, K=K, K2=K2)
        tiles = ((M + 255) // 256) * (N // 256)
        assert p["tiles"] == tiles
        T, S = p["tail_tiles"], p["ranges"]
        if T == 0:
            assert S == 0
            continue
        cases += 1
        nssu = K // 64 + K2 // 64
        assert p["kernel"] == "w4" and tiles > NCU and T == tiles % NCU and 2 <= S <= 8 and T * S <= 2 * NCU and S <= nssu
        if not force:
            assert nssu // S >= 8, "the cost model never cuts ranges shorter than 8 K-tiles"
        rs = _ranges(p, nssu)
        assert len(rs) == T * S and sorted(r[3] for r in rs) == list(range(T * S)), "one slot per range"
        assert max(r[3] for r in rs) * 262144 < _WS_BYTES
        cover = {}
        for t, k0, k1, slot in rs:
            assert 0 <= t < T and 0 <= k0 < k1 <= nssu, "empty or out-of-range range"
            cover.setdefault(t, []).append((k0, k1, slot))
        for t in range(T):
            parts = sorted(cover[t])
            assert [a for a, _, _ in parts] == [(j * nssu) // S for j in range(S)] and parts[-1][1] == nssu
            assert all(parts[i][1] == parts[i + 1][0] for i in range(S - 1)), "ranges of a tile must abut"
            assert [s for _, _, s in parts] == [j * T + t for j in range(S)], "the fix-up kernel reads slot j T + t for part j"
    assert cases > 200


_WS_BYTES = 2 * NCU * 262144     # utx_gemm_streamk_workspace_bytes on a 256-CU device


def test_smaller_grids_and_margins():
    _lib.set_option("UTX_GEMM_PERS_GRID", 200)
    p = ops.gemm_plan(13376, 3072, K=12288, K2=64)        # 636 tiles on 200 workgroups: 3 rounds + 36 tiles
    assert p["tail_tiles"] == 636 % 200 and p["ranges"] * p["tail_tiles"] <= 400
    _lib.set_option("UTX_GEMM_PERS_GRID", 0)
    _lib.set_option("UTX_GEMM_STREAMK", 60)               # a margin of 60 K-tiles: K = 3072 (49 K-tiles per tile) can never clear it
    assert ops.gemm_plan(13376, 3072, K=3072, K2=64)["tail_tiles"] == 0
    assert ops.gemm_plan(13376, 3072, K=12288, K2=64)["tail_tiles"] == 124
