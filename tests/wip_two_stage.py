"""Two-stage generator against the fused one (not collected until it has passed on the GPU:
run as  pytest tests/wip_two_stage.py)."""
import numpy as np
import pytest

from common import assert_canon_equal, assert_planes_equal, load_golden
from oracle.pyoracle import OracleBoard

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("waves_per_board")]
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def goenv():
    from elf_amd import goenv as m
    m.load_library()
    assert torch.cuda.is_available()
    return m


def test_generate_two_stage_equals_fused(goenv, monkeypatch):
    """goenv_generate's two stages (rules kernel -> packed samples -> expander on a second stream) write
    exactly what the fused generator writes: a board count that is a multiple of neither 4 (boards per
    expander workgroup) nor 8 (XCDs), a ring depth that is not a multiple of the chunk, several calls in a
    row (the packed buffers are reused across calls), every D4 code, planes only and all three rings."""
    B, T, seed = 45, 37, 5
    monkeypatch.setenv("GOENV_TWO_STAGE", "0")          # read by goenv_create
    fused = goenv.GoEnv(B, seed=seed, max_ply=70)
    monkeypatch.setenv("GOENV_TWO_STAGE", "1")
    staged = goenv.GoEnv(B, seed=seed, max_ply=70)
    monkeypatch.delenv("GOENV_TWO_STAGE")
    assert 1 <= staged.generate_chunk() <= 16 and T % staged.generate_chunk() != 0
    dev = fused.device
    augs = ((torch.arange(B, dtype=torch.int32, device=dev) * 3) % 8).contiguous()
    rings = [(torch.full((T, B, 25, 19, 19), -7.0, device=dev), torch.full((T, B, 361), 9, dtype=torch.uint8, device=dev),
              torch.full((T, B), -9, dtype=torch.int32, device=dev)) for _ in range(2)]
    for r in range(5):
        planes_only = r == 3
        for env, (f, lg, ac) in ((fused, rings[0]), (staged, rings[1])):
            if planes_only:
                env.generate(T, f, aug_codes=augs)
            else:
                env.generate(T, f, lg, ac, aug_codes=augs)
        torch.cuda.synchronize()
        for x, y in zip(rings[0], rings[1]):
            assert torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x,
                               y.view(torch.int32) if y.dtype == torch.float32 else y), r
    sa, sb = fused.get_state(), staged.get_state()
    for k in sa:
        x, y = (sa[k].view(torch.int16), sb[k].view(torch.int16)) if sa[k].dtype == torch.uint16 else (sa[k], sb[k])
        assert torch.equal(x, y), k
    assert fused.stats() == staged.stats() and fused.stats()[goenv.ST_GAMES] > B
    fused.close()
    staged.close()

