"""Engine: one bpmi_ctx (one GPU, one stream) plus byte-level helpers.

This is the only module that talks to libbpmi.so.  Everything above it (the
reference-shaped Python call surface in pippenger/, utils/, innerproduct/,
rangeproofs/) goes through an Engine; nothing here or above computes EC or bulk
scalar arithmetic on the CPU.
"""
import ctypes
import os
import threading

from . import _native

Q = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
P_FIELD = 2**256 - 2**32 - 977


class HostBuffer:
    """Page-locked host memory owned by an Engine (bpmi_host_alloc): `view` is a writable memoryview over it (recv_into it,
    slice-assign into it), `ptr` its address.  Uploads from it need no staging copy."""

    def __init__(self, engine, nbytes):
        self.engine = engine
        self.nbytes = nbytes
        p = ctypes.c_void_p()
        engine._ck(engine.lib.bpmi_host_alloc(engine.ctx, nbytes, ctypes.byref(p)))
        self.ptr = p.value
        self.view = memoryview((ctypes.c_char * nbytes).from_address(self.ptr)).cast("B")

    def __len__(self):
        return self.nbytes

    def free(self):
        if self.ptr:
            self.view.release()
            self.engine.lib.bpmi_host_free(self.engine.ctx, self.ptr)
            self.ptr = None


class EngineError(RuntimeError):
    pass


class DeviceBuffer:
    """hipMalloc'd bytes owned by an Engine."""

    def __init__(self, engine, nbytes):
        self.engine = engine
        self.nbytes = nbytes
        p = ctypes.c_void_p()
        engine._ck(engine.lib.bpmi_malloc(engine.ctx, nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def upload(self, data, offset=0):
        assert offset + len(data) <= self.nbytes
        self.engine._ck(self.engine.lib.bpmi_upload(self.engine.ctx, self.ptr + offset, bytes(data), len(data)))
        return self

    def download(self, nbytes=None, offset=0):
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = ctypes.create_string_buffer(nbytes)
        self.engine._ck(self.engine.lib.bpmi_download(self.engine.ctx, out, self.ptr + offset, nbytes))
        return out.raw

    def free(self):
        if self.ptr:
            self.engine.lib.bpmi_free(self.engine.ctx, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    def __init__(self, device=None, stream=None):
        self.lib = _native.load()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) % max(self.lib.bpmi_device_count(), 1)
        self.device = device
        self.ctx = self.lib.bpmi_ctx_create(device, stream)
        if not self.ctx:
            raise EngineError("bpmi_ctx_create failed: %s" % self.lib.bpmi_last_error(None).decode())
        # A/B experiments on code that creates its own engines (the bench's batches in flight): BPMI_OPTIONS="name=value,name=value"
        for kv in filter(None, os.environ.get("BPMI_OPTIONS", "").split(",")):
            name, _, value = kv.partition("=")
            self.set_option(name.strip(), int(value))

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.bpmi_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise EngineError("libbpmi error %d: %s" % (rc, self.lib.bpmi_last_error(self.ctx).decode()))

    def set_option(self, name, value):
        self._ck(self.lib.bpmi_set_option(self.ctx, name.encode(), int(value)))

    def sync(self):
        self._ck(self.lib.bpmi_sync(self.ctx))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def host_alloc(self, nbytes):
        return HostBuffer(self, nbytes)

    def upload(self, data):
        return DeviceBuffer(self, max(len(data), 16)).upload(data)

    # ---- byte-level operations (wire format of include/bpmi.h) ----
    def msm_bytes(self, pts, scalars, n):
        out = ctypes.create_string_buffer(64)
        self._ck(self.lib.bpmi_msm(self.ctx, pts, scalars, n, out))
        return out.raw

    def msm2_bytes(self, pts0, scalars0, n0, pts1, scalars1, n1):
        """Two independent MSMs overlapped on the engine's two lanes -> (64 bytes, 64 bytes)."""
        o0, o1 = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
        self._ck(self.lib.bpmi_msm2(self.ctx, pts0, scalars0, n0, o0, pts1, scalars1, n1, o1))
        return o0.raw, o1.raw

    def msm_dev(self, d_pts, d_scalars, n):
        out = ctypes.create_string_buffer(64)
        self._ck(self.lib.bpmi_msm_dev(self.ctx, _ptr(d_pts), _ptr(d_scalars), n, out))
        return out.raw

    def msm_dev_enqueue(self, slot, d_pts, d_scalars, n):
        """Queue one MSM (device pointers, n <= 2^23) in pending slot 0 or 1 and return at once."""
        self._ck(self.lib.bpmi_msm_dev_enqueue(self.ctx, slot, _ptr(d_pts), _ptr(d_scalars), n))

    def msm_finish(self, slot):
        """Wait for the MSM of `slot` only and return its 64-byte result."""
        out = ctypes.create_string_buffer(64)
        self._ck(self.lib.bpmi_msm_finish(self.ctx, slot, out))
        return out.raw

    def msm_geometry(self, n, pipelined=False):
        """What an MSM of n pairs runs as under the current options (bpmi_msm_geometry): a dict with the kernel family, window bits,
        windows (and how many are one bit wider), buckets, chunk length, slices and pairs per slice.  No GPU work."""
        g = (ctypes.c_uint32 * 8)()
        self._ck(self.lib.bpmi_msm_geometry(self.ctx, n, 1 if pipelined else 0, g))
        return {"kernel": ("pipeline", "small", "mid")[g[0]] if n else None, "window_bits": g[1], "windows": g[2], "wide_windows": g[3],
                "buckets": g[4], "chunk": g[5], "slices": g[6], "pairs_per_slice": g[7]}

    def ec_mul_batch_bytes(self, pts, scalars, n):
        out = ctypes.create_string_buffer(64 * n)
        self._ck(self.lib.bpmi_ec_mul_batch(self.ctx, pts, scalars, n, out))
        return out.raw

    def ec_lincomb2_batch_bytes(self, p1, p2, k1, k2, n):
        out = ctypes.create_string_buffer(64 * n)
        self._ck(self.lib.bpmi_ec_lincomb2_batch(self.ctx, p1, p2, k1, k2, n, out))
        return out.raw

    def ec_sum_bytes(self, pts, n):
        out = ctypes.create_string_buffer(64)
        self._ck(self.lib.bpmi_ec_sum(self.ctx, pts, n, out))
        return out.raw

    def ec_sum_dev(self, d_pts, n):
        out = ctypes.create_string_buffer(64)
        self._ck(self.lib.bpmi_ec_sum_dev(self.ctx, _ptr(d_pts), n, out))
        return out.raw

    def ec_decompress_batch_bytes(self, comp, n):
        """n x 33-byte SEC1 compressed points -> (n x 64-byte wire points, n validity flags)."""
        out = ctypes.create_string_buffer(64 * n)
        ok = ctypes.create_string_buffer(max(n, 1))
        self._ck(self.lib.bpmi_ec_decompress_batch(self.ctx, comp, n, out, ok))
        return out.raw, ok.raw[:n]

    def sc_dot_bytes(self, a, b, n):
        out = ctypes.create_string_buffer(32)
        self._ck(self.lib.bpmi_sc_dot(self.ctx, a, b, n, out))
        return out.raw

    def sc_fold_bytes(self, lo, hi, x, y, n):
        out = ctypes.create_string_buffer(32 * n)
        self._ck(self.lib.bpmi_sc_fold(self.ctx, lo, hi, x, y, n, out))
        return out.raw

    def sc_svector_bytes(self, xs, xinvs, k, a, b, scale=None):
        """(sa, sb) of bpmi_sc_svector as packed bytes, 2^k scalars each."""
        n = 1 << k
        sa, sb = ctypes.create_string_buffer(32 * n), ctypes.create_string_buffer(32 * n)
        self._ck(self.lib.bpmi_sc_svector(self.ctx, xs, xinvs, k, sc_bytes(a), sc_bytes(b), scale, sa, sb))
        return sa.raw, sb.raw

    def ipa_verify_dev(self, d_g, d_h, n, xs, xinvs, a, b, extra_pts, extra_scalars, n_extra, d_hscale=None):
        """The 64-byte value of Verifier2's combined check over device-resident generators (identity = accept)."""
        out = ctypes.create_string_buffer(64)
        k = n.bit_length() - 1
        self._ck(self.lib.bpmi_ipa_verify_dev(self.ctx, _ptr(d_g), _ptr(d_h), None if d_hscale is None else _ptr(d_hscale), n, xs, xinvs, k,
                                              sc_bytes(a), sc_bytes(b), extra_pts, extra_scalars, n_extra, out))
        return out.raw

    # ---- IPA prover state ----
    def ipa_create(self, g, h, a, b, n, u, h_scale=None):
        st = ctypes.c_void_p()
        if h_scale is None:
            self._ck(self.lib.bpmi_ipa_create(self.ctx, g, h, a, b, n, u, ctypes.byref(st)))
        else:
            self._ck(self.lib.bpmi_ipa_create_scaled(self.ctx, g, h, a, b, n, u, h_scale, ctypes.byref(st)))
        return IpaState(self, st.value)

    def ipa_create_dev(self, d_g, d_h, d_a, d_b, n, u):
        st = ctypes.c_void_p()
        self._ck(self.lib.bpmi_ipa_create_dev(self.ctx, _ptr(d_g), _ptr(d_h), _ptr(d_a), _ptr(d_b), n, u, ctypes.byref(st)))
        return IpaState(self, st.value)

    # ---- profiling ----
    def profile(self, enable=True):
        """True / 1: HIP events around every stage; 2: around the dominant stage only; False: off."""
        self._ck(self.lib.bpmi_profile(self.ctx, int(enable)))

    def profile_reset(self):
        self._ck(self.lib.bpmi_profile_reset(self.ctx))

    def profile_read(self):
        ms = (ctypes.c_double * _native.NSTAGES)()
        calls = (ctypes.c_uint64 * _native.NSTAGES)()
        self._ck(self.lib.bpmi_profile_read(self.ctx, ms, calls))
        return {self.lib.bpmi_profile_stage_name(i).decode(): (ms[i], calls[i]) for i in range(_native.NSTAGES)}


class IpaState:
    """One FastNIProver2 run on the device, split at the Fiat-Shamir edge."""

    def __init__(self, engine, handle):
        self.engine = engine
        self.handle = handle

    def __len__(self):
        return self.engine.lib.bpmi_ipa_len(self.handle)

    def round_LR(self):
        L = ctypes.create_string_buffer(64)
        R = ctypes.create_string_buffer(64)
        self.engine._ck(self.engine.lib.bpmi_ipa_round_LR(self.handle, L, R))
        return L.raw, R.raw

    def fold(self, x, xinv):
        self.engine._ck(self.engine.lib.bpmi_ipa_fold(self.handle, sc_bytes(x), sc_bytes(xinv)))

    def prove_rounds(self, digest):
        """Every remaining round in one native call (bpmi_ipa_prove_rounds): the transcript `digest` so far ->
        (transcript after the last round, [x as int], [L as 64 bytes], [R as 64 bytes])."""
        k = max(1, len(self).bit_length())
        cap = len(digest) + 256 * k
        out, out_len, rounds = ctypes.create_string_buffer(cap), ctypes.c_uint64(0), ctypes.c_uint32(0)
        xs, Ls, Rs = ctypes.create_string_buffer(32 * k), ctypes.create_string_buffer(64 * k), ctypes.create_string_buffer(64 * k)
        self.engine._ck(self.engine.lib.bpmi_ipa_prove_rounds(self.handle, digest, len(digest), out, cap, ctypes.byref(out_len), xs, Ls, Rs, k,
                                                              ctypes.byref(rounds)))
        r = rounds.value
        return (out.raw[:out_len.value], [int.from_bytes(xs.raw[32 * i: 32 * i + 32], "little") for i in range(r)],
                [Ls.raw[64 * i: 64 * i + 64] for i in range(r)], [Rs.raw[64 * i: 64 * i + 64] for i in range(r)])

    def finish(self):
        a = ctypes.create_string_buffer(32)
        b = ctypes.create_string_buffer(32)
        self.engine._ck(self.engine.lib.bpmi_ipa_finish(self.handle, a, b))
        return int.from_bytes(a.raw, "little"), int.from_bytes(b.raw, "little")

    def export(self):
        """(g, h, a, b) of the current round as packed bytes: len(self) points / scalars each."""
        m = len(self)
        g, h = ctypes.create_string_buffer(64 * m), ctypes.create_string_buffer(64 * m)
        a, b = ctypes.create_string_buffer(32 * m), ctypes.create_string_buffer(32 * m)
        self.engine._ck(self.engine.lib.bpmi_ipa_export(self.handle, g, h, a, b))
        return g.raw, h.raw, a.raw, b.raw

    def close(self):
        if self.handle:
            self.engine.lib.bpmi_ipa_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ptr(x):
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):  # torch tensor on the engine's device
        return x.data_ptr()
    return int(x)


def sc_bytes(k):
    return (int(k) % Q).to_bytes(32, "little")


_default = None
_default_lock = threading.Lock()


def default_engine():
    """Process-wide engine (like the reference's PipSECP256k1 singleton it is stateless
    between calls).  Raises if libbpmi.so or the GPU is missing."""
    global _default
    with _default_lock:
        if _default is None:
            _default = Engine()
        return _default


def set_default_engine(engine):
    global _default
    with _default_lock:
        _default = engine
