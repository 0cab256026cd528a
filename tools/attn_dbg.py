"""Dev harness (GPU): the attention op hook on a tiny case against numpy, element by element (used while the bf16 kernels were written). Not a test."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_amd import ops
rng = np.random.default_rng(0)
n, H, S = 1, 1, 17
D = 64
def ref_attn(qkv):
    q, k, v = [qkv[..., i*D:(i+1)*D].astype(np.float64) for i in range(3)]
    s = q @ k.transpose(0, 2, 1) * 0.125
    e = np.exp(s - s.max(-1, keepdims=True)); a = e / e.sum(-1, keepdims=True)
    return a @ v
# test A: q = 0 -> uniform P; V = small ints
qkv = np.zeros((n, S, 3 * D), np.float32)
qkv[..., 2*D:] = rng.integers(-4, 5, (n, S, D))
out = ops.attention(qkv, H, 1); ref = ref_attn(qkv)
print("A (q=0) max err", np.abs(out - ref).max())
if np.abs(out-ref).max() > 0.05:
    print(" out[0,0,:8]", out[0,0,:8], "\n ref[0,0,:8]", ref[0,0,:8])
    # which V rows explain out?  solve out = w @ V
    V = qkv[0, :, 2*D:].astype(np.float64)
    w, *_ = np.linalg.lstsq(V.T, out[0, 0].astype(np.float64), rcond=None)
    print(" weights over keys for query 0:", np.round(w, 3))
# test B: one-hot attention: q_i = 64*e_{i mod 64}, k_j = e_{j} scaled -> query i attends to key (i mod 64)... S=17 <64 fine
qkv = np.zeros((n, S, 3 * D), np.float32)
for i in range(S):
    qkv[0, i, i] = 64.0           # q_i = 64 e_i
    qkv[0, i, D + (i * 5) % S] = 16.0   # k_i = 16 e_{5i mod S}  -> query j matches key i where 5i mod S == j
qkv[..., 2*D:] = rng.integers(-4, 5, (n, S, D))
out = ops.attention(qkv, H, 1); ref = ref_attn(qkv)
print("B (one-hot) max err", np.abs(out - ref).max())
if np.abs(out-ref).max() > 0.05:
    V = qkv[0, :, 2*D:]
    for i in range(S):
        m = [j for j in range(S) if np.allclose(out[0, i], V[j], atol=0.1)]
        want = [j for j in range(S) if (5 * j) % S == i]
        print(f"  query {i}: out matches V rows {m}, want {want}")
