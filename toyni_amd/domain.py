"""Mirror of the caller of the hot path, `BabyBearDomain` (src/math/domain.rs:10-175), restricted to what
reaches the GPU: fft / ifft with `use_gpu` set.  The coset pre/post scaling that the reference does in a
serial host loop (src/math/domain.rs:154-174) runs on the device here."""
import numpy as np

from . import ntt as _ntt

P = 2013265921


class BabyBearDomain:
    def __init__(self, size: int):
        assert size > 0 and size & (size - 1) == 0, "Domain size must be power of 2"  # src/math/domain.rs:21
        self.size = size
        self.log_size = size.bit_length() - 1
        self.shift = 1
        self.use_gpu = False

    def get_coset(self, shift: int) -> "BabyBearDomain":  # src/math/domain.rs:34-42
        d = BabyBearDomain(self.size)
        d.shift = int(shift) % P
        d.use_gpu = self.use_gpu
        return d

    def with_gpu(self, use_gpu: bool) -> "BabyBearDomain":  # src/math/domain.rs:45-48
        self.use_gpu = use_gpu
        return self

    def _require_gpu(self):
        if not self.use_gpu:
            raise NotImplementedError("toyni_amd ships the GPU backend only; the CPU transform is the reference's src/ntt.rs")
        if not _ntt.gpu_available():
            raise _ntt.ToyniError("GPU not available", 10004)

    def fft(self, coeffs) -> np.ndarray:
        """src/math/domain.rs:107-123: zero-pad to size, scale by shift^i, NTT."""
        self._require_gpu()
        c = np.asarray(coeffs, dtype=np.uint64)
        assert c.size <= self.size
        # the zero padding of :109 is implied on the device: only the coefficients cross PCIe on the way in
        return _ntt.get_or_create_ctx(self.size).lde_host(c, shift=self.shift)

    def ifft(self, evals) -> np.ndarray:
        """src/math/domain.rs:85-102: INTT, then scale by shift^-i."""
        self._require_gpu()
        values = np.array(evals, dtype=np.uint64, copy=True)
        assert values.size == self.size, "Evaluation count must match domain size"  # src/math/domain.rs:86
        _ntt.get_or_create_ctx(self.size).run_host(values, inverse=True, shift=self.shift)
        return values

    def fft_ext(self, coeffs4) -> np.ndarray:
        """src/math/domain.rs:134-151: the base transform of every coordinate -- the AoS vector goes through the interleaved passes as it is."""
        return self._transform_ext(coeffs4, inverse=False)

    def ifft_ext(self, evals4) -> np.ndarray:
        return self._transform_ext(evals4, inverse=True)

    def _transform_ext(self, vals4, inverse: bool) -> np.ndarray:
        self._require_gpu()
        v = np.asarray(vals4, dtype=np.uint64).reshape(-1, 4)
        ctx = _ntt.get_or_create_ctx(self.size)
        if not inverse:
            assert v.shape[0] <= self.size
            # zero padding of :136-137 implied on the device; the reference's de-interleave / four transforms / recombine (:140-151) is one call
            # on the AoS vector (interleaved pass kernels: nothing is de-interleaved)
            return ctx.lde_ext_host(v, shift=self.shift)
        assert v.shape[0] == self.size
        vals = v.copy()
        ctx.run_host_ext(vals.reshape(-1), inverse=True, shift=self.shift)
        return vals
