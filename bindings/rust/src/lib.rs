//! UNVERIFIED (no Rust toolchain in the build image).  Drop-in module tree for callers of density-rs 0.16.6 that routes the
//! encode/decode path to libdensity_hip.so.  Paths, names and signatures follow the reference:
//!   density_rs::algorithms::chameleon::chameleon::Chameleon::{encode, decode}   (src/algorithms/chameleon/chameleon.rs:45-53)
//!   density_rs::codec::codec::Codec::safe_encode_buffer_size                    (src/codec/codec.rs:18-21)
//!   density_rs::errors::{encode_error::EncodeError, decode_error::DecodeError}  (src/errors/*.rs)

#[link(name = "density_hip")]
extern "C" {
    // include/density_hip.h section 1 == src/algorithms/chameleon/chameleon.rs:70-83 (and cheetah.rs:105-118, lion.rs:193-206)
    fn chameleon_encode(input: *const u8, input_size: usize, output: *mut u8, output_size: usize) -> usize;
    fn chameleon_decode(input: *const u8, input_size: usize, output: *mut u8, output_size: usize) -> usize;
    fn chameleon_safe_encode_buffer_size(size: usize) -> usize;
    fn cheetah_encode(input: *const u8, input_size: usize, output: *mut u8, output_size: usize) -> usize;
    fn cheetah_decode(input: *const u8, input_size: usize, output: *mut u8, output_size: usize) -> usize;
    fn cheetah_safe_encode_buffer_size(size: usize) -> usize;
    fn lion_encode(input: *const u8, input_size: usize, output: *mut u8, output_size: usize) -> usize;
    fn lion_decode(input: *const u8, input_size: usize, output: *mut u8, output_size: usize) -> usize;
    fn lion_safe_encode_buffer_size(size: usize) -> usize;
}

pub mod errors {
    pub mod encode_error { #[derive(Debug)] pub struct EncodeError {} }
    pub mod decode_error { #[derive(Debug)] pub struct DecodeError {} }
}

pub mod codec {
    pub mod codec {
        pub trait Codec {
            fn safe_encode_buffer_size(size: usize) -> usize;
        }
    }
}

macro_rules! algo {
    ($m:ident, $t:ident, $enc:ident, $dec:ident, $safe:ident) => {
        pub mod $m {
            pub mod $m {
                use crate::codec::codec::Codec;
                use crate::errors::decode_error::DecodeError;
                use crate::errors::encode_error::EncodeError;
                pub struct $t {}
                impl $t {
                    pub fn encode(input: &[u8], output: &mut [u8]) -> Result<usize, EncodeError> {
                        if input.is_empty() { return Ok(0); }
                        let n = unsafe { crate::$enc(input.as_ptr(), input.len(), output.as_mut_ptr(), output.len()) };
                        if n == 0 { Err(EncodeError {}) } else { Ok(n) }
                    }
                    pub fn decode(input: &[u8], output: &mut [u8]) -> Result<usize, DecodeError> {
                        if input.is_empty() { return Ok(0); }
                        let n = unsafe { crate::$dec(input.as_ptr(), input.len(), output.as_mut_ptr(), output.len()) };
                        if n == 0 { Err(DecodeError {}) } else { Ok(n) }
                    }
                }
                impl Codec for $t {
                    fn safe_encode_buffer_size(size: usize) -> usize { unsafe { crate::$safe(size) } }
                }
            }
        }
    };
}

pub mod algorithms {
    algo!(chameleon, Chameleon, chameleon_encode, chameleon_decode, chameleon_safe_encode_buffer_size);
    algo!(cheetah, Cheetah, cheetah_encode, cheetah_decode, cheetah_safe_encode_buffer_size);
    algo!(lion, Lion, lion_encode, lion_decode, lion_safe_encode_buffer_size);
}
