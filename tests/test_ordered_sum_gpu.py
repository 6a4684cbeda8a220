"""The parallel parity scan of csrc/parity_scan.hpp (ramses_amd_ordered_sum_device) against the sequential sum it must
reproduce bit for bit: s <- fl(s + x[i]) in list order (numpy's cumsum adds exactly like that).  It carries the ordered dot
products of the conjugate-gradient solver (signed terms; poisson/phi_fine_cg.f90:98-105,146-153) and rho_fine's multipole
sums (positive terms; tests/test_rho_fine_gpu.py)."""
import ctypes as C
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_sum(gpu_lib, x):
    import torch
    x = np.ascontiguousarray(x, np.float64)
    d_x = torch.from_numpy(x).cuda() if x.size else torch.zeros(1, dtype=torch.float64, device="cuda")
    d_out = torch.full((1,), 123.0, dtype=torch.float64, device="cuda")
    nb = gpu_lib.ramses_amd_ordered_sum_scratch(x.size)
    d_w = torch.zeros(int(nb), dtype=torch.uint8, device="cuda")
    rc = gpu_lib.ramses_amd_ordered_sum_device(C.c_void_p(d_x.data_ptr()), x.size, C.c_void_p(d_out.data_ptr()),
                                               C.c_void_p(d_w.data_ptr()), None)
    assert rc == 0, gpu_lib.ramses_amd_last_error()
    torch.cuda.synchronize()
    return float(d_out.item())


def _seq(x):
    return float(np.cumsum(np.asarray(x, np.float64))[-1]) if len(x) else 0.0


def _same(a, b):
    return np.float64(a).tobytes() == np.float64(b).tobytes()


CASES = {
    "normal_mean_1": lambda r: r.normal(1.0, 1.0, 3_000_000),
    "normal_mean_0": lambda r: r.normal(0.0, 1.0, 400_000),                       # the sum changes sign again and again
    "mostly_positive_products": lambda r: r.normal(0.3, 1.0, 2_000_000) * r.normal(0.3, 1.0, 2_000_000),
    "all_negative": lambda r: -r.random(1_500_000),
    "negative_mean": lambda r: r.normal(-0.7, 1.0, 1_000_000),
    "sixteen_decades_signed": lambda r: r.normal(0.2, 1.0, 1_000_000) * 10.0 ** r.uniform(-12, 4, 1_000_000),
    "positive_sixteen_decades": lambda r: 10.0 ** r.uniform(-12, 4, 1_000_000),
    "exact_ties": lambda r: r.choice([0.5, 1.5, -0.5, 2.5, 1.0, -1.5, 0.25, 3.0], 1_000_000) * 2.0 ** -30 + 0.0,
    "ties_on_a_large_sum": lambda r: np.concatenate([[2.0 ** 30], r.choice([2.0 ** -23, -2.0 ** -23, 3 * 2.0 ** -23, 2.0 ** -22], 600_000)]),
    "leading_zeros": lambda r: np.concatenate([np.zeros(50_000), r.normal(0.5, 1.0, 300_000)]),
    "cancels_to_zero_and_restarts": lambda r: np.concatenate([[1.0, -1.0], np.zeros(20_000), r.random(100_000), [-3.0e4], r.random(100_000)]),
    "falls_through_a_power_of_two": lambda r: np.concatenate([[1024.0], -r.random(400_000) * 1e-2]),
    "huge_then_tiny": lambda r: np.concatenate([[1e300, -1e300, 1e-300], r.normal(0, 1e-290, 10_000), r.normal(1.0, 0.1, 50_000)]),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_scan_equals_the_sequential_sum(gpu_lib, name):
    x = CASES[name](np.random.default_rng(zlib.crc32(name.encode())))
    got, ref = _device_sum(gpu_lib, x), _seq(x)
    assert _same(got, ref), (name, got, ref, got - ref)
    # and it is not the pairwise sum in disguise wherever the two differ
    assert np.isfinite(ref)


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 8191, 8192, 8193, 16384, 100_000])
def test_scan_sizes_around_the_segment_length(gpu_lib, n):
    rng = np.random.default_rng(n + 3)
    x = rng.normal(0.1, 1.0, n)
    assert _same(_device_sum(gpu_lib, x), _seq(x))
    assert _same(_device_sum(gpu_lib, np.zeros(n)), 0.0)


def test_scan_differs_from_the_pairwise_sum_it_replaces(gpu_lib):
    """what the test above would not see if both sums agreed anyway: on this input the sequential and the tree sum differ"""
    x = np.random.default_rng(2).normal(1.0, 1.0, 3_000_000)
    seq, tree = _seq(x), float(np.sum(x))
    assert seq != tree
    assert _same(_device_sum(gpu_lib, x), seq)
