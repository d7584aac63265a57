"""helpers shared by the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import numpy as np


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def pts_as_f32(p):
    """oracle float3 record array -> float32 [...,3]"""
    return p.view(np.float32).reshape(p.shape + (3,))


def mean_records(t):
    """[k,16] uint8 tensor of kde_superpixel -> structured numpy array"""
    from oracle.oracle import SUPERPIXEL
    return host(t).view(SUPERPIXEL).reshape(-1)


def ld_records(t):
    from oracle.oracle import LABEL_DISTANCE
    a = host(t)
    return a.view(LABEL_DISTANCE).reshape(a.shape[:2])
