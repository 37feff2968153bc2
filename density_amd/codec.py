"""Host-side mirror of the reference crate's public interface for the encode/decode path.

Reference surface (density-rs 0.16.6):
    density_rs::algorithms::chameleon::chameleon::Chameleon::{encode, decode}(input: &[u8], output: &mut [u8])
        -> Result<usize, EncodeError | DecodeError>                       (chameleon.rs:45-53)
    likewise cheetah::cheetah::Cheetah (cheetah.rs:57-65) and lion::lion::Lion (lion.rs:74-82)
    <T as codec::codec::Codec>::safe_encode_buffer_size(usize) -> usize   (codec/codec.rs:18-21)

Same names, argument meaning and error behaviour: `encode(input, output)` fills the caller's `output` buffer and
returns the number of bytes written; failure raises EncodeError / DecodeError (the Rust `Err`).  Every call goes
through the C ABI of libdensity_hip.so — the symbols the crate itself exports (chameleon.rs:70-83) — and from
there to the gfx950 kernels.  Like the crate's FFI, an empty input yields 0 bytes.
"""
import ctypes

from . import _lib


class EncodeError(Exception):
    """errors/encode_error.rs"""


class DecodeError(Exception):
    """errors/decode_error.rs"""


def _ro(buf):
    """read-only buffer -> (address, nbytes, keepalive)"""
    if hasattr(buf, "__array_interface__"):
        return buf.__array_interface__["data"][0], buf.nbytes, buf
    mv = memoryview(buf).cast("B")
    n = mv.nbytes
    if n == 0:
        return 0, 0, mv
    if mv.readonly:
        c = (ctypes.c_char * n).from_buffer_copy(mv)
    else:
        c = (ctypes.c_char * n).from_buffer(mv)
    return ctypes.addressof(c), n, c


def _rw(buf):
    if hasattr(buf, "__array_interface__"):
        if buf.__array_interface__["data"][1]:
            raise TypeError("output buffer is read-only")
        return buf.__array_interface__["data"][0], buf.nbytes, buf
    mv = memoryview(buf).cast("B")
    if mv.readonly:
        raise TypeError("output buffer is read-only")
    n = mv.nbytes
    if n == 0:
        return 0, 0, mv
    c = (ctypes.c_char * n).from_buffer(mv)
    return ctypes.addressof(c), n, c


class _Codec:
    NAME = None

    @classmethod
    def safe_encode_buffer_size(cls, size):
        """Codec::safe_encode_buffer_size, codec/codec.rs:18-21"""
        return getattr(_lib.lib(), f"{cls.NAME}_safe_encode_buffer_size")(size)

    @classmethod
    def encode(cls, input, output):
        """{Algo}::encode(input, output) -> bytes written (one reference-format stream)."""
        ia, n, k1 = _ro(input)
        oa, cap, k2 = _rw(output)
        if n == 0:
            return 0
        r = getattr(_lib.lib(), f"{cls.NAME}_encode")(ia, n, oa, cap)
        if r == 0:
            raise EncodeError(_lib.last_error())
        return r

    @classmethod
    def decode(cls, input, output):
        """{Algo}::decode(input, output) -> bytes written."""
        ia, n, k1 = _ro(input)
        oa, cap, k2 = _rw(output)
        if n == 0:
            return 0
        r = getattr(_lib.lib(), f"{cls.NAME}_decode")(ia, n, oa, cap)
        if r == 0:
            raise DecodeError(_lib.last_error())
        return r


class Chameleon(_Codec):
    """algorithms/chameleon/chameleon.rs"""
    NAME = "chameleon"


class Cheetah(_Codec):
    """algorithms/cheetah/cheetah.rs"""
    NAME = "cheetah"


class Lion(_Codec):
    """algorithms/lion/lion.rs"""
    NAME = "lion"


BY_NAME = {"chameleon": Chameleon, "cheetah": Cheetah, "lion": Lion}
