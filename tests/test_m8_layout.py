"""Host restatement of the LDS placements of the p = 7 stage A (exahype_amd/csrc/exa_dg_m8.hpp, M8Geo): the claims the kernel's comments make about bank
conflicts -- checked here with the MI355X LDS rules (ds_read_b64: 32-lane groups over 64 banks of 4 B = residues mod 32 doubles; ds_write_b64: 16-lane groups,
residues mod 16 doubles).  No GPU, no library call: pure arithmetic on the formulas (kept in step with the header by tests/test_stage_a_variants.py, which runs
the kernel against the oracle)."""
from collections import Counter

N = 8


def ypair(j):
    return ((j & 1) << 1) | (j >> 1)


def _read_conflicts(bases, stride, pairs):
    """worst multiplicity of a bank over the 32-lane groups of one derive round: lane rows (k, k + 1) hold the node pairs pairs[k], pairs[k + 1]"""
    worst = 1
    for half in range(2):
        for mirror in range(2):
            addr = []
            for j in (2 * half, 2 * half + 1):
                n = pairs[j]
                n = 7 - n if mirror else n
                addr += [b + n * stride for b in bases]
            worst = max(worst, max(Counter(x % 32 for x in set(addr)).values()))
    return worst


def test_r5_layout_variant_is_conflict_free():
    """EXA_M8_LAYOUT=1 (measured, not adopted: profiles/r05_m8_kernel.txt 3a): strides 8 / 70 / 560."""
    PY, PX = 8, 70
    slots = {}
    for a in range(8):
        for b in range(8):
            for c in range(8):
                g = ((a >> 2) << 3) | ((b >> 2) << 2) | (c >> 1)
                slots[(g, (6 * a + 8 * b + c) & 31)] = (a, b, c)
    assert len(slots) == 512                                                 # owner_slot is a bijection
    zslot = lambda a, b, c: (2 * (a >> 2) + (b >> 2)) * 128 + c * 16 + 4 * (a & 3) + (b & 3)
    assert {zslot(a, b, c) for a in range(8) for b in range(8) for c in range(8)} == set(range(512))
    for g in range(16):
        q = [(PX * a + PY * b + c) % 32 for a, b, c in (slots[(g, l)] for l in range(32))]
        assert sorted(q) == list(range(32))                                  # owners: 32 distinct banks in Q ...
        assert all(len({x % 16 for x in q[16 * h:16 * h + 16]}) == 16 for h in range(2))      # ... and per 16-lane store group
        assert len({zslot(*slots[(g, l)]) % 32 for l in range(32)}) == 32    # ... and in the compact z array
    for k in range(4):                                                       # y rounds: node pairs in the order 0, 2, 1, 3
        bases = [(k + 4 * (p >> 3)) * PX + (p & 7) for p in range(16)]
        assert _read_conflicts(bases, PY, [ypair(j) for j in range(4)]) == 1
        assert all(len({(b + n * PY) % 16 for b in bases}) == 16 for n in range(8))           # in-place stores
    for w in range(4):                                                       # z rounds
        bases = [(4 * (w >> 1) + (p >> 2)) * PX + (4 * (w & 1) + (p & 3)) * PY for p in range(16)]
        assert _read_conflicts(bases, 1, list(range(4))) == 1


def test_r5_plane_variant_claims():
    """EXA_M8_PLANE=1 (measured, not adopted: profiles/r05_m8_kernel.txt 3b): strides 8 / 68 / 562, wave w owns the nodes (a, b = w, c)."""
    PY, PX, SL = 8, 68, 562
    slot = {}
    for a in range(8):
        for b in range(8):
            for c in range(8):
                slot[b * 64 + (c >> 2) * 32 + ((4 * a + c) & 31)] = (a, b, c)
    assert len(slot) == 512
    for w in range(8):
        assert all(slot[w * 64 + l][1] == w for l in range(64))
        res = [(PX * slot[w * 64 + l][0] + PY * w + slot[w * 64 + l][2]) for l in range(64)]
        assert all(len({x % 32 for x in res[32 * g:32 * g + 32]}) == 32 for g in range(2))
        assert all(len({x % 16 for x in res[16 * g:16 * g + 16]}) == 16 for g in range(4))
        zb = [(p >> 3) * SL + PX * (p & 7) + PY * w for p in range(16)]
        assert _read_conflicts(zb, 1, list(range(4))) == 1
        xb = [(p >> 3) * SL + PY * w + (p & 7) for p in range(16)]
        assert _read_conflicts(xb, PX, [ypair(j) for j in range(4)]) == 2    # 2 of 32 lanes share a bank with another
