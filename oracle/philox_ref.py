"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatement of the Dropout keep mask of the stage-2 training step.

The reference's heads hold ``nn.Dropout(0.5)`` (module2_mixed/my_models.py:110-121, ``refinement_head.net0``); on a CUDA machine
torch draws its mask with the Philox4x32-10 counter-based generator (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy
as 1, 2, 3", SC'11 - the published algorithm restated here).  ``me_dropout_mask_u8`` (millieye_amd/csrc/m2_train.hip) uses the same
generator with its own counter layout: key = the 64-bit seed, counter = (quad index low, quad index high, 0, 0), the four output
words decide elements 4q .. 4q + 3, keep iff (word >> 8) / 2^24 < keep_prob.  Pinned against the known-answer vectors of the
Random123 distribution (tests/test_oracle_golden.py)."""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter [..., 4] uint32, key (k0, k1) -> [..., 4] uint32 (ten rounds)."""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)) & _MASK, p1 & _MASK,
             ((p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)) & _MASK, p0 & _MASK]
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1).astype(np.uint32)


def dropout_mask(seed, keep_prob, count):
    """uint8 [count]: the mask me_dropout_mask_u8(seed, keep_prob, count) writes."""
    quads = (count + 3) // 4
    q = np.arange(quads, dtype=np.uint64)
    ctr = np.zeros((quads, 4), dtype=np.uint32)
    ctr[:, 0] = (q & _MASK).astype(np.uint32)
    ctr[:, 1] = (q >> np.uint64(32)).astype(np.uint32)
    words = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)).reshape(-1)[:count]
    u = (words >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (u < np.float32(keep_prob)).astype(np.uint8)
