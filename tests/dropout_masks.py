"""numpy restatement of the counter-based dropout masks of the fused training decoder kernels (test helper).

The kernels store no masks: both passes regenerate every keep decision from a hash of (device seed, salt, dropout
stream, element index) -- dpft_amd/csrc/decoder_train.hip:45-70 (``drop_hash``, ``drop_ctx``, ``attn_pair_index``) and
dpft_amd/csrc/decoder_train_x.hip:63-78 (``xdrop_scale``).  Replaying the same decisions here lets a test run the ORACLE
with exactly the kernel's masks, i.e. compare dropout > 0 element by element instead of statistically."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _hash(idx, s0, s1):
    x = (idx.astype(np.uint64) ^ np.uint64(s0)) & M32
    x = (x * np.uint64(0xcc9e2d51)) & M32
    x = ((x << np.uint64(15)) | (x >> np.uint64(17))) & M32
    x = (x * np.uint64(0x1b873593)) & M32
    x ^= np.uint64(s1)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85ebca6b)) & M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xc2b2ae35)) & M32
    x ^= x >> np.uint64(16)
    return x


def _ctx(seed: int, salt: int, stream_id: int, p: float):
    seed &= (1 << 64) - 1
    s0 = ((seed & 0xFFFFFFFF) ^ ((salt * 0x9E3779B9) & 0xFFFFFFFF)) & 0xFFFFFFFF
    s1 = ((seed >> 32) + stream_id * 0x7F4A7C15) & 0xFFFFFFFF
    thr = int(np.float32(p) * np.float32(65536.0) + np.float32(0.5))
    return s0, s1, thr


def _keep(idx, which, seed, salt, stream_id, p):
    """1/(1-p) where the 16-bit field ``which`` of hash(idx) >= p * 65536, else 0."""
    s0, s1, thr = _ctx(seed, salt, stream_id, p)
    h = _hash(idx, s0, s1)
    field = (h >> (np.uint64(16) * which.astype(np.uint64))) & np.uint64(0xFFFF)
    return np.where(field >= np.uint64(thr), 1.0 / (1.0 - p), 0.0)


def self_attn_masks(seed: int, salt: int, p: float, V: int, B: int, Q: int, heads: int = 8, C: int = 16):
    """-> att (V,B,heads,Q,Q) keep-scales of the attention probabilities (stream 0: one hash per key pair (k, k+8) of a
    16-key group), d1 (V,B,Q,C) keep-scales of dropout1 (stream 1: one hash per channel pair)."""
    KP = ((Q + 15) >> 4) * 8
    vb = np.arange(V * B, dtype=np.uint64).reshape(V * B, 1, 1, 1)
    h = np.arange(heads, dtype=np.uint64).reshape(1, heads, 1, 1)
    q = np.arange(Q, dtype=np.uint64).reshape(1, 1, Q, 1)
    k = np.arange(Q, dtype=np.uint64).reshape(1, 1, 1, Q)
    idx = (((vb * np.uint64(heads) + h) * np.uint64(Q) + q) * np.uint64(KP) + (k >> np.uint64(4)) * np.uint64(8) + (k & np.uint64(7))) & M32
    att = _keep(idx, np.broadcast_to((k >> np.uint64(3)) & np.uint64(1), idx.shape), seed, salt, 0, p).reshape(V, B, heads, Q, Q)
    row = np.arange(V * B * Q, dtype=np.uint64).reshape(-1, 1)
    c = np.arange(C, dtype=np.uint64).reshape(1, C)
    d1 = _keep((row * np.uint64(8) + (c >> np.uint64(1))) & M32, np.broadcast_to(c & np.uint64(1), (V * B * Q, C)), seed, salt, 1, p)
    return att, d1.reshape(V, B, Q, C)


def xattn_ffn_masks(seed: int, salt: int, p: float, V: int, B: int, Q: int, C: int = 16, d_ffn: int = 32):
    """-> d2 (V,B,Q,C) after the cross attention (stream 2), d3 (V,B,Q,d_ffn) inside the FFN (stream 3), d4 (V,B,Q,C)
    after the FFN (stream 4): element e of a stream -> hash(e >> 1), field e & 1."""
    out = []
    for stream_id, n in ((2, C), (3, d_ffn), (4, C)):
        e = np.arange(V * B * Q * n, dtype=np.uint64)
        out.append(_keep((e >> np.uint64(1)) & M32, e & np.uint64(1), seed, salt, stream_id, p).reshape(V, B, Q, n))
    return out
