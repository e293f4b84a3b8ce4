"""Seeded input generators shared by tools/gen_goldens.py (which feeds them to the imported
reference) and the tests (which feed the same arrays to the oracle and the HIP path).
Pure numpy: inputs never depend on a torch RNG stream ('RNG is not portable')."""
import numpy as np


def latents(seed, B, C, zdim, scale=1.0):
    rs = np.random.RandomState(seed)
    z = (rs.standard_normal((B, zdim)) * scale).astype(np.float32)
    centres = (rs.standard_normal((C, zdim)) * scale).astype(np.float32)
    return z, centres


def clustered_latents(seed, B, C, zdim, n_clusters=10, spread=0.35):
    """Latents with cluster structure (closer to a trained encoder's output than iid noise):
    rows are a cluster centre plus isotropic noise."""
    rs = np.random.RandomState(seed)
    mu = rs.standard_normal((n_clusters, zdim)).astype(np.float32)
    z = (mu[rs.randint(0, n_clusters, B)] + spread * rs.standard_normal((B, zdim))).astype(np.float32)
    c = (mu[rs.randint(0, n_clusters, C)] + spread * rs.standard_normal((C, zdim))).astype(np.float32)
    return z, c


def mask_indices(seed, B, C, N):
    """Batch indices [B x 1] and exemplar indices [C] drawn with replacement from range(N); rows 0..2
    are forced to have 2, 1 and 3 matches so duplicate / multi-match masking is exercised."""
    rs = np.random.RandomState(seed)
    z_idx = rs.randint(0, N, size=(B, 1)).astype(np.int64)
    c_idx = rs.randint(0, N, size=(C,)).astype(np.int64)
    if C >= 8 and B >= 3:
        c_idx[0] = c_idx[5] = z_idx[0, 0]
        c_idx[1] = z_idx[1, 0]
        c_idx[2] = c_idx[3] = c_idx[7] = z_idx[2, 0]
    return z_idx, c_idx


def binary_images(seed, N, D=784, n_classes=10, ink=0.13):
    """MNIST-like synthetic binary images: x_i ~ Bernoulli(P_{i mod 10}), P_k a smooth
    field squashed so mean ink is about `ink`.  Returns float32 {0,1} [N x D]."""
    rs = np.random.RandomState(seed)
    side = int(round(D ** 0.5))
    protos = []
    for _ in range(n_classes):
        f = rs.standard_normal((side, side))
        for _ in range(3):                                   # cheap low-pass: 3x3 box blur x3
            f = (f + np.roll(f, 1, 0) + np.roll(f, -1, 0) + np.roll(f, 1, 1) + np.roll(f, -1, 1)) / 5.0
        f = (f - f.mean()) / (f.std() + 1e-8)
        protos.append(1.0 / (1.0 + np.exp(-(3.0 * f - 2.6))))
    protos = np.stack(protos).reshape(n_classes, -1)[:, :D]
    cls = np.arange(N) % n_classes
    x = (rs.random_sample((N, D)) < protos[cls]).astype(np.float32)
    return x


def gray_images(seed, N, D=784):
    """Un-binarised variant (values k/255) used as exemplar images: the reference encodes exemplars
    from the raw dataset tensor while the batch is binarised (training.py:31 vs BaseModel.py:247)."""
    rs = np.random.RandomState(seed)
    return ((rs.randint(0, 256, size=(N, D)) / 255.0) * (rs.random_sample((N, D)) < 0.2)).astype(np.float32)


def g23_inputs(B, C, N, D, zdim, seed=95):
    """G23 (single_conv, 3 x 64 x 64 at a batch that switches the pixel-image operators on): N structured colour images (a coarse 8 x 8
    random field per channel, nearest-upsampled, plus pixel noise; values (k + 0.5) / 256 as utils/load_data/base_load_data.py:36
    produces), a batch of B of them slightly perturbed, C distinct candidate rows, eps."""
    rs = np.random.RandomState(seed)
    side = int(round((D // 3) ** 0.5))
    coarse = rs.randint(32, 224, (N, 3, 8, 8)).astype(np.float32)
    img = np.repeat(np.repeat(coarse, side // 8, axis=2), side // 8, axis=3) + rs.randint(-24, 25, (N, 3, side, side))
    data = ((np.clip(img, 0, 255).astype(np.int64) + 0.5) / 256).astype(np.float32).reshape(N, D)
    bidx = rs.choice(N, size=(B, 1), replace=False).astype(np.int64)
    x = np.clip(data[bidx[:, 0]] + rs.randint(-6, 7, (B, D)).astype(np.float32) / 256, 0.5 / 256, 255.5 / 256).astype(np.float32)
    cand = rs.choice(N, size=C, replace=False).astype(np.int64)
    eps = rs.standard_normal((B, zdim)).astype(np.float32)
    return data, x, bidx, cand, eps
