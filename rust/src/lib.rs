//! milagro_bls-compatible facade over the C ABI of `libmbls_hip.so` (`include/mbls.h`).
//!
//! **Source only -- never compiled or run**: the build image has no Rust toolchain. It is written against the
//! public API of sigp/milagro_bls v1.5.1 (`src/lib.rs:17-22`): every re-exported type keeps its name, its methods, their argument
//! meaning and their error behaviour; the bodies call the GPU library instead of the `amcl` crate. The same ABI is exercised end to
//! end by the Python mirror (`milagro_bls_amd/api.py`) and the C++ mirror (`include/milagro_bls.hpp`), which the GPU tests run.
//!
//! Differences a caller can observe:
//! * `point` fields hold serialized points (amcl's `ECP`/`ECP2` do not exist here): `PublicKey.point` / `AggregatePublicKey.point`
//!   are the 96-byte uncompressed form, `Signature.point` / `AggregateSignature.point` the 96-byte compressed form.
//! * `BLSCurve` (`pub use amcl::bls381 as BLSCurve`, `src/lib.rs:17`) is a passthrough of the amcl crate and is not provided.
//! * The batch entry points (`batch::*`, `KeyTable`) are additions: they are what the GPU exists for.
#![allow(clippy::missing_safety_doc)]

extern crate rand;
extern crate zeroize;

use rand::Rng;
use std::os::raw::{c_char, c_int, c_void};

/// `mbls_scalar_source` of include/mbls.h
type MblsScalarSource = unsafe extern "C" fn(user: *mut c_void, out: *mut u64, count: u64);
use std::sync::Once;
use zeroize::Zeroize;

pub const G1_BYTES: usize = 48; // reference src/lib.rs:20
pub const G2_BYTES: usize = 96;
pub const SECRET_KEY_BYTES: usize = 32;

/// The `AmclError` variants the reference uses (`amcl::errors::AmclError`, re-exported at `src/amcl_utils.rs:11`).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum AmclError {
    AggregateEmptyPoints,
    InvalidSecretKeySize,
    InvalidSecretKeyRange,
    InvalidPoint,
    InvalidG1Size,
    InvalidG2Size,
}

// ------------------------------------------------------------------------------------------------ the C ABI (include/mbls.h)
#[repr(C)]
pub struct MblsCtx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct MblsKeyTable {
    _p: [u8; 0],
}
extern "C" {
    fn mbls_ctx_create(out: *mut *mut MblsCtx, device_id: c_int) -> c_int;
    fn mbls_last_error(ctx: *mut MblsCtx) -> *const c_char;
    fn mbls_pk_from_bytes(ctx: *mut MblsCtx, bytes: *const u8, len: usize, pk_out: *mut u8) -> c_int;
    fn mbls_pk_from_bytes_unchecked(ctx: *mut MblsCtx, bytes: *const u8, len: usize, pk_out: *mut u8) -> c_int;
    fn mbls_pk_from_uncompressed_bytes(ctx: *mut MblsCtx, bytes: *const u8, len: usize, pk_out: *mut u8) -> c_int;
    fn mbls_pk_as_bytes(ctx: *mut MblsCtx, pk: *const u8, out: *mut u8) -> c_int;
    fn mbls_pk_key_validate(ctx: *mut MblsCtx, pk: *const u8) -> c_int;
    fn mbls_pk_from_secret_key(ctx: *mut MblsCtx, sk: *const u8, sk_len: usize, pk_out: *mut u8) -> c_int;
    fn mbls_sig_from_bytes(ctx: *mut MblsCtx, bytes: *const u8, len: usize, sig_out: *mut u8) -> c_int;
    fn mbls_sign(ctx: *mut MblsCtx, msg: *const u8, msg_len: usize, sk: *const u8, sk_len: usize, sig_out: *mut u8) -> c_int;
    fn mbls_verify(ctx: *mut MblsCtx, sig: *const u8, msg: *const u8, msg_len: usize, pk: *const u8) -> c_int;
    fn mbls_aggregate_public_keys(ctx: *mut MblsCtx, pks96: *const u8, n: usize, apk_out: *mut u8) -> c_int;
    fn mbls_aggregate_public_key_add(ctx: *mut MblsCtx, a: *const u8, b: *const u8, out: *mut u8) -> c_int;
    fn mbls_aggregate_signature_add(ctx: *mut MblsCtx, a: *const u8, b: *const u8, out: *mut u8) -> c_int;
    fn mbls_aggregate_signatures_batch(ctx: *mut MblsCtx, sigs96: *const u8, offsets: *const u32, n: u64, k: u32, out96: *mut u8, errs: *mut u8) -> c_int;
    fn mbls_fast_aggregate_verify(ctx: *mut MblsCtx, sig: *const u8, msg: *const u8, msg_len: usize, pks96: *const u8, n_pks: usize) -> c_int;
    fn mbls_fast_aggregate_verify_pre_aggregated(ctx: *mut MblsCtx, sig: *const u8, msg: *const u8, msg_len: usize, apk: *const u8) -> c_int;
    fn mbls_aggregate_verify(ctx: *mut MblsCtx, sig: *const u8, msgs: *const u8, msg_lens: *const usize, n_msgs: usize, pks96: *const u8, n_pks: usize) -> c_int;
    fn mbls_verify_multiple_aggregate_signatures(ctx: *mut MblsCtx, sigs96: *const u8, apks96: *const u8, msgs: *const u8, msg_len: u32, msg_offsets: *const u64,
                                                 rands: *const u64, n: usize) -> c_int;
    fn mbls_verify_multiple_aggregate_signatures_rng(ctx: *mut MblsCtx, sigs96: *const u8, apks96: *const u8, msgs: *const u8, msg_len: u32, msg_offsets: *const u64,
                                                     n: usize, draw: MblsScalarSource, user: *mut c_void) -> c_int;
    fn mbls_sig_check_batch(ctx: *mut MblsCtx, in96: *const u8, n: u64, errs: *mut u8, in_g2: *mut u8) -> c_int;
    fn mbls_fast_aggregate_verify_batch(ctx: *mut MblsCtx, sigs: *const u8, msgs: *const u8, msg_len: u32, msg_offsets: *const u64, pks: *const u8, pk_format: c_int,
                                        pk_offsets: *const u32, n: u64, k: u32, results: *mut u8, status: *mut u32) -> c_int;
    fn mbls_aggregate_verify_batch(ctx: *mut MblsCtx, sigs96: *const u8, msgs: *const u8, msg_len: u32, msg_offsets: *const u64, pks96: *const u8,
                                   pair_offsets: *const u32, k: u32, n: u64, results: *mut u8, status: *mut u32) -> c_int;
    fn mbls_verify_batch(ctx: *mut MblsCtx, sigs: *const u8, msgs: *const u8, msg_len: u32, msg_offsets: *const u64, pks: *const u8, pk_format: c_int, n: u64,
                         results: *mut u8, status: *mut u32) -> c_int;
    fn mbls_keytable_create(ctx: *mut MblsCtx, capacity_hint: u64, out: *mut *mut MblsKeyTable) -> c_int;
    fn mbls_keytable_destroy(t: *mut MblsKeyTable);
    fn mbls_keytable_size(t: *const MblsKeyTable) -> u64;
    fn mbls_keytable_append(t: *mut MblsKeyTable, pks: *const u8, pk_format: c_int, validate: c_int, n: u64, first_index: *mut u64, errs: *mut u8) -> c_int;
    fn mbls_keytable_get(t: *mut MblsKeyTable, first_index: u64, n: u64, pks96: *mut u8, errs: *mut u8) -> c_int;
    fn mbls_fast_aggregate_verify_batch_indexed(ctx: *mut MblsCtx, t: *const MblsKeyTable, sigs: *const u8, msgs: *const u8, msg_len: u32, msg_offsets: *const u64,
                                                key_idx: *const u32, offsets: *const u32, n: u64, k: u32, results: *mut u8, status: *mut u32) -> c_int;
}
#[repr(C)]
pub struct MblsMulti {
    _private: [u8; 0],
}
extern "C" {
    fn mbls_multi_create(out: *mut *mut MblsMulti, device_ids: *const c_int, n_devices: c_int) -> c_int;
    fn mbls_multi_destroy(m: *mut MblsMulti);
    fn mbls_multi_device_count(m: *const MblsMulti) -> c_int;
    fn mbls_multi_fast_aggregate_verify_batch(m: *mut MblsMulti, sigs: *const u8, msgs: *const u8, msg_len: u32, msg_offsets: *const u64, pks: *const u8,
                                              pk_format: c_int, pk_offsets: *const u32, n: u64, k: u32, results: *mut u8, status: *mut u32) -> c_int;
    fn mbls_multi_rccl_active(m: *const MblsMulti) -> c_int;
    fn mbls_multi_fast_aggregate_verify_bitmap(m: *mut MblsMulti, sigs: *const u8, msgs: *const u8, msg_len: u32, msg_offsets: *const u64, pks: *const u8,
                                               pk_format: c_int, pk_offsets: *const u32, n: u64, k: u32, bitmap: *mut u64, status: *mut u32) -> c_int;
    fn mbls_multi_verify_multiple_aggregate_signatures(m: *mut MblsMulti, sigs96: *const u8, apks96: *const u8, msgs: *const u8, msg_len: u32,
                                                       msg_offsets: *const u64, rands: *const u64, n: usize) -> c_int;
    fn mbls_multi_verify_multiple_aggregate_signatures_rng(m: *mut MblsMulti, sigs96: *const u8, apks96: *const u8, msgs: *const u8, msg_len: u32,
                                                           msg_offsets: *const u64, n: usize, draw: MblsScalarSource, user: *mut c_void) -> c_int;
}
const PK_COMPRESSED: c_int = 0;
const PK_UNCOMPRESSED: c_int = 1;

struct CtxPtr(*mut MblsCtx);
unsafe impl Send for CtxPtr {}
unsafe impl Sync for CtxPtr {} // every ABI entry takes the context's lock (include/mbls.h)
static INIT: Once = Once::new();
static mut CTX: CtxPtr = CtxPtr(std::ptr::null_mut());

/// The process-wide context (GPU 0, or `MBLS_DEVICE`). There is no CPU fallback: without an MI355X this panics, like a missing
/// dynamic library would.
fn ctx() -> *mut MblsCtx {
    unsafe {
        INIT.call_once(|| {
            let dev = std::env::var("MBLS_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
            let mut p: *mut MblsCtx = std::ptr::null_mut();
            let rc = mbls_ctx_create(&mut p, dev);
            if rc != 0 {
                panic!("mbls_ctx_create failed ({}): no MI355X / HIP device -- libmbls_hip has no CPU fallback", rc);
            }
            CTX = CtxPtr(p);
        });
        CTX.0
    }
}

fn err(code: c_int) -> AmclError {
    match code {
        1 => AmclError::InvalidG1Size,
        2 => AmclError::InvalidG2Size,
        3 => AmclError::InvalidPoint,
        4 => AmclError::AggregateEmptyPoints,
        5 => AmclError::InvalidSecretKeySize,
        6 => AmclError::InvalidSecretKeyRange,
        _ => {
            let msg = unsafe { std::ffi::CStr::from_ptr(mbls_last_error(ctx())) };
            panic!("mbls device error {}: {}", code, msg.to_string_lossy())
        }
    }
}
fn check(code: c_int) -> Result<(), AmclError> {
    if code == 0 {
        Ok(())
    } else {
        Err(err(code))
    }
}

// ------------------------------------------------------------------------------------------------ host-side SHA-256 / HKDF (KeyGenerate)
mod hkdf {
    const K: [u32; 64] = [
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
        0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
        0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2,
    ];
    pub fn sha256(msg: &[u8]) -> [u8; 32] {
        let mut h: [u32; 8] = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19];
        let mut m = msg.to_vec();
        m.push(0x80);
        while m.len() % 64 != 56 {
            m.push(0);
        }
        m.extend_from_slice(&((msg.len() as u64) * 8).to_be_bytes());
        for blk in m.chunks(64) {
            let mut w = [0u32; 64];
            for i in 0..16 {
                w[i] = u32::from_be_bytes([blk[4 * i], blk[4 * i + 1], blk[4 * i + 2], blk[4 * i + 3]]);
            }
            for i in 16..64 {
                let s0 = w[i - 15].rotate_right(7) ^ w[i - 15].rotate_right(18) ^ (w[i - 15] >> 3);
                let s1 = w[i - 2].rotate_right(17) ^ w[i - 2].rotate_right(19) ^ (w[i - 2] >> 10);
                w[i] = w[i - 16].wrapping_add(s0).wrapping_add(w[i - 7]).wrapping_add(s1);
            }
            let (mut a, mut b, mut c, mut d, mut e, mut f, mut g, mut hh) = (h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
            for i in 0..64 {
                let t1 = hh.wrapping_add(e.rotate_right(6) ^ e.rotate_right(11) ^ e.rotate_right(25)).wrapping_add((e & f) ^ (!e & g)).wrapping_add(K[i]).wrapping_add(w[i]);
                let t2 = (a.rotate_right(2) ^ a.rotate_right(13) ^ a.rotate_right(22)).wrapping_add((a & b) ^ (a & c) ^ (b & c));
                hh = g; g = f; f = e; e = d.wrapping_add(t1); d = c; c = b; b = a; a = t1.wrapping_add(t2);
            }
            for (x, y) in h.iter_mut().zip([a, b, c, d, e, f, g, hh].iter()) {
                *x = x.wrapping_add(*y);
            }
        }
        let mut out = [0u8; 32];
        for i in 0..8 {
            out[4 * i..4 * i + 4].copy_from_slice(&h[i].to_be_bytes());
        }
        out
    }
    pub fn hmac(key: &[u8], msg: &[u8]) -> [u8; 32] {
        let mut k = if key.len() > 64 { sha256(key).to_vec() } else { key.to_vec() };
        k.resize(64, 0);
        let mut i: Vec<u8> = k.iter().map(|b| b ^ 0x36).collect();
        i.extend_from_slice(msg);
        let mut o: Vec<u8> = k.iter().map(|b| b ^ 0x5c).collect();
        o.extend_from_slice(&sha256(&i));
        sha256(&o)
    }
    /// OS2IP(48 bytes) mod r as 32 big-endian bytes (bitwise long division; host-only, once per key).
    pub fn mod_r(okm: &[u8]) -> [u8; 32] {
        const R: [u32; 9] = [0x0000_0001, 0xffff_ffff, 0xfffe_5bfe, 0x53bd_a402, 0x09a1_d805, 0x3339_d808, 0x299d_7d48, 0x73ed_a753, 0];
        let mut a = [0u32; 9];
        for bit in 0..okm.len() * 8 {
            for j in (1..9).rev() {
                a[j] = (a[j] << 1) | (a[j - 1] >> 31);
            }
            a[0] = (a[0] << 1) | (((okm[bit >> 3] >> (7 - (bit & 7))) & 1) as u32);
            let mut ge = true;
            for j in (0..9).rev() {
                if a[j] != R[j] {
                    ge = a[j] > R[j];
                    break;
                }
            }
            if ge {
                let mut br = 0u64;
                for j in 0..9 {
                    let d = (a[j] as u64).wrapping_sub(R[j] as u64).wrapping_sub(br);
                    a[j] = d as u32;
                    br = (d >> 63) & 1;
                }
            }
        }
        let mut o = [0u8; 32];
        for j in 0..8 {
            o[4 * j..4 * j + 4].copy_from_slice(&a[7 - j].to_be_bytes());
        }
        o
    }
}

// ------------------------------------------------------------------------------------------------ keys (reference src/keys.rs)
/// Domain for key generation (`src/keys.rs:24`).
pub const KEY_SALT: &[u8] = b"BLS-SIG-KEYGEN-SALT-";
/// L = ceil((3 * ceil(log2(r))) / 16) = 48 (`src/keys.rs:26`).
pub const L: u8 = 48;
const CURVE_ORDER_BE: [u8; 32] = [
    0x73, 0xed, 0xa7, 0x53, 0x29, 0x9d, 0x7d, 0x48, 0x33, 0x39, 0xd8, 0x08, 0x09, 0xa1, 0xd8, 0x05, 0x53, 0xbd, 0xa4, 0x02, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0xff,
    0xff, 0xff, 0x00, 0x00, 0x00, 0x01,
];

/// A BLS secret key (`src/keys.rs:28-113`): 32 big-endian bytes, host-only; the scalar goes to the GPU only for signing.
#[derive(Clone)]
pub struct SecretKey {
    x: [u8; SECRET_KEY_BYTES],
}
impl SecretKey {
    /// `src/keys.rs:36-39`
    pub fn random<R: Rng + ?Sized>(rng: &mut R) -> Self {
        let ikm: [u8; 32] = rng.gen();
        Self::key_generate(&ikm, &[]).unwrap() // will only error if ikm < 32 bytes
    }
    /// KeyGenerate, `src/keys.rs:45-77`.
    pub fn key_generate(ikm: &[u8], key_info: &[u8]) -> Result<Self, AmclError> {
        if ikm.len() < 32 {
            return Err(AmclError::InvalidSecretKeySize);
        }
        let mut sk = [0u8; 32];
        let mut salt = KEY_SALT.to_vec();
        while sk.iter().all(|b| *b == 0) {
            salt = hkdf::sha256(&salt).to_vec(); // salt = H(salt)
            let mut ikm0 = ikm.to_vec();
            ikm0.push(0);
            let prk = hkdf::hmac(&salt, &ikm0); // PRK = HKDF-Extract(salt, IKM || I2OSP(0, 1))
            ikm0.zeroize();
            let mut info = key_info.to_vec();
            info.extend_from_slice(&[0, L]); // key_info || I2OSP(L, 2)
            let mut okm: Vec<u8> = Vec::with_capacity(64);
            let mut t: Vec<u8> = Vec::new();
            let mut ctr = 1u8;
            while okm.len() < L as usize {
                let mut m = t.clone();
                m.extend_from_slice(&info);
                m.push(ctr);
                ctr += 1;
                t = hkdf::hmac(&prk, &m).to_vec();
                okm.extend_from_slice(&t);
            }
            sk = hkdf::mod_r(&okm[..L as usize]); // SK = OS2IP(OKM) mod r
            okm.zeroize();
        }
        Ok(Self { x: sk })
    }
    /// `src/keys.rs:80-82`; error cases pinned by the reference tests at `src/keys.rs:285-297`.
    pub fn from_bytes(input: &[u8]) -> Result<SecretKey, AmclError> {
        if input.len() != SECRET_KEY_BYTES {
            return Err(AmclError::InvalidSecretKeySize);
        }
        if input.iter().all(|b| *b == 0) || input >= &CURVE_ORDER_BE[..] {
            return Err(AmclError::InvalidSecretKeyRange);
        }
        let mut x = [0u8; 32];
        x.copy_from_slice(input);
        Ok(Self { x })
    }
    /// `src/keys.rs:85-87`
    pub fn as_bytes(&self) -> [u8; SECRET_KEY_BYTES] {
        self.x
    }
}
impl PartialEq for SecretKey {
    fn eq(&self, other: &SecretKey) -> bool {
        self.as_bytes() == other.as_bytes()
    }
}
impl Eq for SecretKey {}
impl Drop for SecretKey {
    fn drop(&mut self) {
        self.x.zeroize(); // src/keys.rs:109-113
    }
}

/// A BLS public key (`src/keys.rs:116-187`). `point` = the 96-byte uncompressed form (x || y; infinity = 0x40 || 0..).
#[derive(Clone, PartialEq, Eq, Debug)]
pub struct PublicKey {
    pub point: [u8; 96],
}
impl PublicKey {
    /// `src/keys.rs:124-137`
    pub fn from_secret_key(sk: &SecretKey) -> Self {
        let mut p = [0u8; 96];
        let b = sk.as_bytes();
        check(unsafe { mbls_pk_from_secret_key(ctx(), b.as_ptr(), b.len(), p.as_mut_ptr()) }).expect("a SecretKey is always in range");
        PublicKey { point: p }
    }
    /// Compressed decode + KeyValidate, `src/keys.rs:140-147`.
    pub fn from_bytes(bytes: &[u8]) -> Result<PublicKey, AmclError> {
        let mut p = [0u8; 96];
        check(unsafe { mbls_pk_from_bytes(ctx(), bytes.as_ptr(), bytes.len(), p.as_mut_ptr()) })?;
        Ok(PublicKey { point: p })
    }
    /// `src/keys.rs:150-155`
    pub fn from_bytes_unchecked(bytes: &[u8]) -> Result<PublicKey, AmclError> {
        let mut p = [0u8; 96];
        check(unsafe { mbls_pk_from_bytes_unchecked(ctx(), bytes.as_ptr(), bytes.len(), p.as_mut_ptr()) })?;
        Ok(PublicKey { point: p })
    }
    /// `src/keys.rs:158-160`
    pub fn as_bytes(&self) -> [u8; G1_BYTES] {
        let mut o = [0u8; G1_BYTES];
        check(unsafe { mbls_pk_as_bytes(ctx(), self.point.as_ptr(), o.as_mut_ptr()) }).expect("a PublicKey always holds a decoded point");
        o
    }
    /// `src/keys.rs:163-165`
    pub fn as_uncompressed_bytes(&self) -> [u8; 2 * G1_BYTES] {
        self.point
    }
    /// `src/keys.rs:170-175`
    pub fn from_uncompressed_bytes(bytes: &[u8]) -> Result<PublicKey, AmclError> {
        let mut p = [0u8; 96];
        check(unsafe { mbls_pk_from_uncompressed_bytes(ctx(), bytes.as_ptr(), bytes.len(), p.as_mut_ptr()) })?;
        Ok(PublicKey { point: p })
    }
    /// KeyValidate, `src/keys.rs:181-186`.
    pub fn key_validate(&self) -> bool {
        unsafe { mbls_pk_key_validate(ctx(), self.point.as_ptr()) == 1 }
    }
}

/// `src/keys.rs:189-204`
#[derive(Clone, PartialEq, Eq)]
pub struct Keypair {
    pub sk: SecretKey,
    pub pk: PublicKey,
}
impl Keypair {
    pub fn random<R: Rng + ?Sized>(rng: &mut R) -> Self {
        let sk = SecretKey::random(rng);
        let pk = PublicKey::from_secret_key(&sk);
        Keypair { sk, pk }
    }
}

// ------------------------------------------------------------------------------------------------ signatures (reference src/signature.rs)
/// `src/signature.rs:9-51`. `point` = the 96-byte compressed form.
#[derive(Clone, PartialEq, Eq, Debug)]
pub struct Signature {
    pub point: [u8; G2_BYTES],
}
impl Signature {
    /// `src/signature.rs:17-21`
    pub fn new(msg: &[u8], sk: &SecretKey) -> Self {
        let mut s = [0u8; G2_BYTES];
        let b = sk.as_bytes();
        check(unsafe { mbls_sign(ctx(), msg.as_ptr(), msg.len(), b.as_ptr(), b.len(), s.as_mut_ptr()) }).expect("a SecretKey is always in range");
        Signature { point: s }
    }
    /// CoreVerify, `src/signature.rs:27-40`.
    pub fn verify(&self, msg: &[u8], pk: &PublicKey) -> bool {
        unsafe { mbls_verify(ctx(), self.point.as_ptr(), msg.as_ptr(), msg.len(), pk.point.as_ptr()) == 1 }
    }
    /// `src/signature.rs:43-46`
    pub fn from_bytes(bytes: &[u8]) -> Result<Signature, AmclError> {
        let mut s = [0u8; G2_BYTES];
        check(unsafe { mbls_sig_from_bytes(ctx(), bytes.as_ptr(), bytes.len(), s.as_mut_ptr()) })?;
        Ok(Signature { point: s })
    }
    /// `src/signature.rs:49-51`
    pub fn as_bytes(&self) -> [u8; G2_BYTES] {
        self.point
    }
}

// ------------------------------------------------------------------------------------------------ aggregates (reference src/aggregates.rs)
/// `src/aggregates.rs:17-78`
#[derive(Clone, PartialEq, Eq, Debug)]
pub struct AggregatePublicKey {
    pub point: [u8; 96],
}
impl AggregatePublicKey {
    /// `src/aggregates.rs:29-39`
    pub fn aggregate(public_keys: &[&PublicKey]) -> Result<Self, AmclError> {
        if public_keys.is_empty() {
            return Err(AmclError::AggregateEmptyPoints);
        }
        let flat: Vec<u8> = public_keys.iter().flat_map(|k| k.point.iter().copied()).collect();
        let mut p = [0u8; 96];
        check(unsafe { mbls_aggregate_public_keys(ctx(), flat.as_ptr(), public_keys.len(), p.as_mut_ptr()) })?;
        Ok(Self { point: p })
    }
    /// `src/aggregates.rs:46-56`
    pub fn into_aggregate(public_keys: &[PublicKey]) -> Result<Self, AmclError> {
        let refs: Vec<&PublicKey> = public_keys.iter().collect();
        Self::aggregate(&refs)
    }
    /// `src/aggregates.rs:61-63`
    pub fn from_public_key(public_key: &PublicKey) -> Self {
        Self { point: public_key.point }
    }
    /// `src/aggregates.rs:68-70`
    pub fn add(&mut self, public_key: &PublicKey) {
        let mut o = [0u8; 96];
        check(unsafe { mbls_aggregate_public_key_add(ctx(), self.point.as_ptr(), public_key.point.as_ptr(), o.as_mut_ptr()) }).expect("decoded points");
        self.point = o;
    }
    /// `src/aggregates.rs:73-77`
    pub fn add_aggregate(&mut self, aggregate_public_key: &AggregatePublicKey) {
        let mut o = [0u8; 96];
        check(unsafe { mbls_aggregate_public_key_add(ctx(), self.point.as_ptr(), aggregate_public_key.point.as_ptr(), o.as_mut_ptr()) }).expect("decoded points");
        self.point = o;
    }
}

/// `src/aggregates.rs:83-334`
#[derive(Clone, PartialEq, Eq, Debug)]
pub struct AggregateSignature {
    pub point: [u8; G2_BYTES],
}
impl Default for AggregateSignature {
    fn default() -> Self {
        Self::new()
    }
}
impl AggregateSignature {
    /// The point at infinity, `src/aggregates.rs:93-95`.
    pub fn new() -> Self {
        let mut p = [0u8; G2_BYTES];
        p[0] = 0xC0;
        Self { point: p }
    }
    /// `src/aggregates.rs:100-106` -- one batched launch (segmented G2 sum) instead of one addition per signature.
    pub fn aggregate(signatures: &[&Signature]) -> Self {
        if signatures.is_empty() {
            return Self::new();
        }
        let flat: Vec<u8> = signatures.iter().flat_map(|s| s.point.iter().copied()).collect();
        let mut out = [0u8; G2_BYTES];
        let mut e = 0u8;
        check(unsafe { mbls_aggregate_signatures_batch(ctx(), flat.as_ptr(), std::ptr::null(), 1, u32::try_from(signatures.len()).expect("signature counts are 32-bit"), out.as_mut_ptr(), &mut e) })
            .and_then(|_| check(e as c_int))
            .expect("Signature objects hold decodable points");
        Self { point: out }
    }
    /// `src/aggregates.rs:109-111`
    pub fn from_signature(signature: &Signature) -> Self {
        Self { point: signature.point }
    }
    /// `src/aggregates.rs:114-116`
    pub fn add(&mut self, signature: &Signature) {
        let mut o = [0u8; G2_BYTES];
        check(unsafe { mbls_aggregate_signature_add(ctx(), self.point.as_ptr(), signature.point.as_ptr(), o.as_mut_ptr()) }).expect("decodable points");
        self.point = o;
    }
    /// `src/aggregates.rs:122-124`
    pub fn add_aggregate(&mut self, aggregate_signature: &AggregateSignature) {
        let mut o = [0u8; G2_BYTES];
        check(unsafe { mbls_aggregate_signature_add(ctx(), self.point.as_ptr(), aggregate_signature.point.as_ptr(), o.as_mut_ptr()) }).expect("decodable points");
        self.point = o;
    }
    /// AggregateVerify, `src/aggregates.rs:130-170`.
    pub fn aggregate_verify(&self, msgs: &[&[u8]], public_keys: &[&PublicKey]) -> bool {
        let flat_m: Vec<u8> = msgs.iter().flat_map(|m| m.iter().copied()).collect();
        let lens: Vec<usize> = msgs.iter().map(|m| m.len()).collect();
        let flat_p: Vec<u8> = public_keys.iter().flat_map(|k| k.point.iter().copied()).collect();
        unsafe { mbls_aggregate_verify(ctx(), self.point.as_ptr(), flat_m.as_ptr(), lens.as_ptr(), msgs.len(), flat_p.as_ptr(), public_keys.len()) == 1 }
    }
    /// FastAggregateVerify, `src/aggregates.rs:177-215`.
    pub fn fast_aggregate_verify(&self, msg: &[u8], public_keys: &[&PublicKey]) -> bool {
        let flat: Vec<u8> = public_keys.iter().flat_map(|k| k.point.iter().copied()).collect();
        unsafe { mbls_fast_aggregate_verify(ctx(), self.point.as_ptr(), msg.as_ptr(), msg.len(), flat.as_ptr(), public_keys.len()) == 1 }
    }
    /// `src/aggregates.rs:223-253`
    pub fn fast_aggregate_verify_pre_aggregated(&self, msg: &[u8], aggregate_public_key: &AggregatePublicKey) -> bool {
        unsafe { mbls_fast_aggregate_verify_pre_aggregated(ctx(), self.point.as_ptr(), msg.as_ptr(), msg.len(), aggregate_public_key.point.as_ptr()) == 1 }
    }
    /// `src/aggregates.rs:261-316`. The blinding scalars are drawn from `rng` exactly as the reference does (8 random bytes,
    /// big-endian i64, absolute value, retry on zero, `:280-287`) -- and in the reference's ORDER: its loop tests set i's signature for
    /// the subgroup (`:272-275`) before it draws `rand[i]` and returns at the first signature outside G2, so a rejected batch leaves the
    /// caller's generator where the reference would. `mbls_verify_multiple_aggregate_signatures_rng` tests the signatures first and asks
    /// for the scalars of the sets before the first bad one only. (The iterator is collected first: the reference stops pulling from it
    /// at the rejected set.)
    pub fn verify_multiple_aggregate_signatures<'a, R, I>(rng: &mut R, signature_sets: I) -> bool
    where
        R: Rng + ?Sized,
        I: Iterator<Item = (&'a AggregateSignature, &'a AggregatePublicKey, &'a [u8])>,
    {
        let sets: Vec<(&AggregateSignature, &AggregatePublicKey, &[u8])> = signature_sets.collect();
        let n = sets.len();
        if n == 0 {
            return true; // e(infinity, -G1) = 1
        }
        let (mut sigs, mut apks, mut msgs) = (Vec::new(), Vec::new(), Vec::new());
        let mut moff: Vec<u64> = vec![0]; // messages of any length each (`&[u8]` per set): one buffer + an offset table
        for (s, a, m) in &sets {
            sigs.extend_from_slice(&s.point);
            apks.extend_from_slice(&a.point);
            msgs.extend_from_slice(m);
            moff.push(msgs.len() as u64);
        }
        // `draw` of mbls_verify_multiple_aggregate_signatures_rng: the scalars of the sets in front of the first bad signature, drawn as `:280-287` does
        let mut st = DrawState { rng, panic: None };
        let ok = unsafe {
            mbls_verify_multiple_aggregate_signatures_rng(ctx(), sigs.as_ptr(), apks.as_ptr(), msgs.as_ptr(), 0, moff.as_ptr(), n, draw_scalars::<R>,
                                                          &mut st as *mut DrawState<R> as *mut c_void) == 1
        };
        if let Some(payload) = st.panic.take() {
            std::panic::resume_unwind(payload);
        }
        ok
    }
    /// `src/aggregates.rs:319-322`
    pub fn from_bytes(bytes: &[u8]) -> Result<AggregateSignature, AmclError> {
        let mut s = [0u8; G2_BYTES];
        check(unsafe { mbls_sig_from_bytes(ctx(), bytes.as_ptr(), bytes.len(), s.as_mut_ptr()) })?;
        Ok(Self { point: s })
    }
    /// `src/aggregates.rs:325-327`
    pub fn as_bytes(&self) -> [u8; G2_BYTES] {
        self.point
    }
}

// The `draw` of mbls_[multi_]verify_multiple_aggregate_signatures_rng: the scalars of the sets in front of the first bad signature, drawn as `:280-287` does.
// A panic must not unwind through the C frames of the library (the context mutex is held and GPU work is in flight): it is caught HERE, the scalars are
// zeroed so that the batch fails closed (a zero scalar is rejected), and the payload is raised again once the FFI call has returned -- what the C++ and
// Python mirrors do with an exception in their generator.
struct DrawState<'r, R: Rng + ?Sized> {
    rng: &'r mut R,
    panic: Option<Box<dyn std::any::Any + Send + 'static>>,
}
unsafe extern "C" fn draw_scalars<R: Rng + ?Sized>(user: *mut c_void, out: *mut u64, count: u64) {
    let st: &mut DrawState<R> = &mut *(user as *mut DrawState<R>);
    let out = std::slice::from_raw_parts_mut(out, count as usize);
    let r = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| {
        for o in out.iter_mut() {
            let mut rand = 0u64;
            while rand == 0 {
                let mut rand_bytes = [0u8; 8];
                st.rng.fill(&mut rand_bytes);
                rand = i64::from_be_bytes(rand_bytes).wrapping_abs() as u64;
            }
            *o = rand;
        }
    }));
    if let Err(payload) = r {
        for o in out.iter_mut() {
            *o = 0;
        }
        st.panic = Some(payload);
    }
}

// ------------------------------------------------------------------------------------------------ additions: the batch path
/// What the GPU exists for: many independent verifications per call (not part of the reference's API).
pub mod batch {
    use super::*;
    /// n x `AggregateSignature::fast_aggregate_verify`: item i = (signatures[i], messages[i], its `k` keys). One bool per item.
    pub fn fast_aggregate_verify(signatures: &[AggregateSignature], messages: &[[u8; 32]], public_keys: &[Vec<&PublicKey>]) -> Vec<bool> {
        let n = signatures.len();
        assert!(messages.len() == n && public_keys.len() == n);
        let sigs: Vec<u8> = signatures.iter().flat_map(|s| s.point.iter().copied()).collect();
        let msgs: Vec<u8> = messages.iter().flat_map(|m| m.iter().copied()).collect();
        let mut offsets: Vec<u32> = Vec::with_capacity(n + 1);
        let mut pks: Vec<u8> = Vec::new();
        offsets.push(0);
        for set in public_keys {
            for k in set {
                pks.extend_from_slice(&k.point);
            }
            offsets.push(u32::try_from(pks.len() / 96).expect("key indices are 32-bit"));
        }
        let mut res = vec![0u8; n];
        let rc = unsafe {
            mbls_fast_aggregate_verify_batch(ctx(), sigs.as_ptr(), msgs.as_ptr(), 32, std::ptr::null(), pks.as_ptr(), PK_UNCOMPRESSED, offsets.as_ptr(), n as u64, 0, res.as_mut_ptr(), std::ptr::null_mut())
        };
        if rc != 0 {
            err(rc);
        }
        res.into_iter().map(|b| b == 1).collect()
    }
    /// n x `AggregateSignature::aggregate_verify` (`src/aggregates.rs:130-170`): item i = (signatures[i], its messages, its keys), as many
    /// messages as keys per item (an item where they differ, or with none, is false like the reference's early return).
    pub fn aggregate_verify(signatures: &[AggregateSignature], messages: &[Vec<&[u8]>], public_keys: &[Vec<&PublicKey>]) -> Vec<bool> {
        let n = signatures.len();
        assert!(messages.len() == n && public_keys.len() == n);
        let sigs: Vec<u8> = signatures.iter().flat_map(|s| s.point.iter().copied()).collect();
        let mut pair_off: Vec<u32> = vec![0];
        let mut msg_off: Vec<u64> = vec![0];
        let (mut msgs, mut pks): (Vec<u8>, Vec<u8>) = (Vec::new(), Vec::new());
        let mut mismatch = vec![false; n];
        for i in 0..n {
            if messages[i].len() != public_keys[i].len() {
                mismatch[i] = true; // src/aggregates.rs:131-133: the item is false; it enters the batch without pairs
            } else {
                for (m, k) in messages[i].iter().zip(public_keys[i].iter()) {
                    msgs.extend_from_slice(m);
                    msg_off.push(msgs.len() as u64);
                    pks.extend_from_slice(&k.point);
                }
            }
            pair_off.push(u32::try_from(pks.len() / 96).expect("pair indices are 32-bit"));
        }
        let mut res = vec![0u8; n];
        let rc = unsafe {
            mbls_aggregate_verify_batch(ctx(), sigs.as_ptr(), msgs.as_ptr(), 0, msg_off.as_ptr(), pks.as_ptr(), pair_off.as_ptr(), 0, n as u64, res.as_mut_ptr(), std::ptr::null_mut())
        };
        if rc != 0 {
            err(rc);
        }
        res.into_iter().zip(mismatch).map(|(b, bad)| b == 1 && !bad).collect()
    }
    /// n x `Signature::verify`.
    pub fn verify(signatures: &[Signature], messages: &[[u8; 32]], public_keys: &[&PublicKey]) -> Vec<bool> {
        let n = signatures.len();
        assert!(messages.len() == n && public_keys.len() == n);
        let sigs: Vec<u8> = signatures.iter().flat_map(|s| s.point.iter().copied()).collect();
        let msgs: Vec<u8> = messages.iter().flat_map(|m| m.iter().copied()).collect();
        let pks: Vec<u8> = public_keys.iter().flat_map(|k| k.point.iter().copied()).collect();
        let mut res = vec![0u8; n];
        let rc = unsafe { mbls_verify_batch(ctx(), sigs.as_ptr(), msgs.as_ptr(), 32, std::ptr::null(), pks.as_ptr(), PK_UNCOMPRESSED, n as u64, res.as_mut_ptr(), std::ptr::null_mut()) };
        if rc != 0 {
            err(rc);
        }
        res.into_iter().map(|b| b == 1).collect()
    }
}

/// Several GPUs behind one handle (`mbls_multi_*`): items are independent (`src/aggregates.rs:177-215` keeps no state between calls), so a
/// batch is cut into contiguous shards, one per device, each staged and verified by its own host thread inside the library.
pub struct MultiGpu {
    h: *mut MblsMulti,
}
unsafe impl Send for MultiGpu {}
unsafe impl Sync for MultiGpu {}
impl MultiGpu {
    pub fn new(device_ids: &[i32]) -> Self {
        let mut h: *mut MblsMulti = std::ptr::null_mut();
        let rc = unsafe { mbls_multi_create(&mut h, device_ids.as_ptr(), device_ids.len() as c_int) };
        if rc != 0 {
            err(rc);
        }
        MultiGpu { h }
    }
    pub fn devices(&self) -> usize {
        unsafe { mbls_multi_device_count(self.h) as usize }
    }
    /// n x `AggregateSignature::fast_aggregate_verify` over all devices; messages of any length each.
    pub fn fast_aggregate_verify(&self, signatures: &[AggregateSignature], messages: &[&[u8]], public_keys: &[Vec<&PublicKey>]) -> Vec<bool> {
        let n = signatures.len();
        assert!(messages.len() == n && public_keys.len() == n);
        let sigs: Vec<u8> = signatures.iter().flat_map(|s| s.point.iter().copied()).collect();
        let mut msgs: Vec<u8> = Vec::new();
        let mut moff: Vec<u64> = vec![0];
        for m in messages {
            msgs.extend_from_slice(m);
            moff.push(msgs.len() as u64);
        }
        let mut offsets: Vec<u32> = vec![0];
        let mut pks: Vec<u8> = Vec::new();
        for set in public_keys {
            for k in set {
                pks.extend_from_slice(&k.point);
            }
            offsets.push(u32::try_from(pks.len() / 96).expect("key indices are 32-bit"));
        }
        let mut res = vec![0u8; n];
        let rc = unsafe {
            mbls_multi_fast_aggregate_verify_batch(self.h, sigs.as_ptr(), msgs.as_ptr(), 0, moff.as_ptr(), pks.as_ptr(), PK_UNCOMPRESSED, offsets.as_ptr(), n as u64, 0,
                                                   res.as_mut_ptr(), std::ptr::null_mut())
        };
        if rc != 0 {
            err(rc);
        }
        res.into_iter().map(|b| b == 1).collect()
    }
    /// Whether the handle's exchange steps (the accept bitmap below, `verify_multiple`'s partial records) run as RCCL all-gathers between the
    /// devices -- over xGMI on an MI355X node -- or, where RCCL could not set up a communicator, through host memory.
    pub fn rccl_active(&self) -> bool {
        unsafe { mbls_multi_rccl_active(self.h) == 1 }
    }
    /// The same verifications with the results as one packed accept bitmap (bit i % 64 of word i / 64 = item i) that every device of the
    /// handle ends up holding: each device packs its shard's bits, the words are all-gathered between the devices. Returns the words.
    pub fn fast_aggregate_verify_bitmap(&self, signatures: &[AggregateSignature], messages: &[&[u8]], public_keys: &[Vec<&PublicKey>]) -> Vec<u64> {
        let n = signatures.len();
        assert!(messages.len() == n && public_keys.len() == n);
        let sigs: Vec<u8> = signatures.iter().flat_map(|s| s.point.iter().copied()).collect();
        let mut msgs: Vec<u8> = Vec::new();
        let mut moff: Vec<u64> = vec![0];
        for m in messages {
            msgs.extend_from_slice(m);
            moff.push(msgs.len() as u64);
        }
        let mut offsets: Vec<u32> = vec![0];
        let mut pks: Vec<u8> = Vec::new();
        for set in public_keys {
            for k in set {
                pks.extend_from_slice(&k.point);
            }
            offsets.push(u32::try_from(pks.len() / 96).expect("key indices are 32-bit"));
        }
        let mut words = vec![0u64; (n + 63) / 64];
        let rc = unsafe {
            mbls_multi_fast_aggregate_verify_bitmap(self.h, sigs.as_ptr(), msgs.as_ptr(), 0, moff.as_ptr(), pks.as_ptr(), PK_UNCOMPRESSED, offsets.as_ptr(), n as u64, 0,
                                                    words.as_mut_ptr(), std::ptr::null_mut())
        };
        if rc != 0 {
            err(rc);
        }
        words
    }
    /// `AggregateSignature::verify_multiple_aggregate_signatures` (`src/aggregates.rs:261-316`) with the sets cut into one shard per device:
    /// every device runs its sets up to its Miller product and signature sum, the first device joins the records and finishes. Same bool.
    pub fn verify_multiple_aggregate_signatures<'a, R, I>(&self, rng: &mut R, signature_sets: I) -> bool
    where
        R: Rng + ?Sized,
        I: Iterator<Item = (&'a AggregateSignature, &'a AggregatePublicKey, &'a [u8])>,
    {
        let sets: Vec<(&AggregateSignature, &AggregatePublicKey, &[u8])> = signature_sets.collect();
        let n = sets.len();
        if n == 0 {
            return true;
        }
        let (mut sigs, mut apks, mut msgs) = (Vec::new(), Vec::new(), Vec::new());
        let mut moff: Vec<u64> = vec![0];
        for (s, a, m) in &sets {
            sigs.extend_from_slice(&s.point);
            apks.extend_from_slice(&a.point);
            msgs.extend_from_slice(m);
            moff.push(msgs.len() as u64);
        }
        // one call, the reference's RNG order (`src/aggregates.rs:272-287`): every device tests its shard's signatures first, the scalars are asked for once
        let mut st = DrawState { rng, panic: None };
        let ok = unsafe {
            mbls_multi_verify_multiple_aggregate_signatures_rng(self.h, sigs.as_ptr(), apks.as_ptr(), msgs.as_ptr(), 0, moff.as_ptr(), n, draw_scalars::<R>,
                                                                &mut st as *mut DrawState<R> as *mut c_void) == 1
        };
        if let Some(payload) = st.panic.take() {
            std::panic::resume_unwind(payload);
        }
        ok
    }
}
impl Drop for MultiGpu {
    fn drop(&mut self) {
        unsafe { mbls_multi_destroy(self.h) }
    }
}

/// Decoded public keys resident in GPU memory: decode (and KeyValidate) once, then verify by index -- the device-side analogue of
/// holding `PublicKey` objects and passing `&[&PublicKey]` (`src/aggregates.rs:177`).
pub struct KeyTable {
    h: *mut MblsKeyTable,
}
unsafe impl Send for KeyTable {}
unsafe impl Sync for KeyTable {}
impl KeyTable {
    pub fn new(capacity_hint: u64) -> Self {
        let mut h: *mut MblsKeyTable = std::ptr::null_mut();
        let rc = unsafe { mbls_keytable_create(ctx(), capacity_hint, &mut h) };
        if rc != 0 {
            err(rc);
        }
        KeyTable { h }
    }
    pub fn len(&self) -> u64 {
        unsafe { mbls_keytable_size(self.h) }
    }
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }
    /// Appends compressed keys through `PublicKey::from_bytes` (decode + KeyValidate). Returns the index of the first new entry and
    /// one `Result` per key; a rejected key still occupies its index as an invalid entry (items naming it are rejected).
    pub fn append_compressed(&mut self, keys: &[[u8; G1_BYTES]]) -> (u64, Vec<Result<(), AmclError>>) {
        let flat: Vec<u8> = keys.iter().flat_map(|k| k.iter().copied()).collect();
        let mut first = 0u64;
        let mut errs = vec![0u8; keys.len()];
        let rc = unsafe { mbls_keytable_append(self.h, flat.as_ptr(), PK_COMPRESSED, 1, keys.len() as u64, &mut first, errs.as_mut_ptr()) };
        if rc != 0 {
            err(rc);
        }
        (first, errs.into_iter().map(|e| check(e as c_int)).collect())
    }
    /// Appends already-decoded keys (no re-validation, like `PublicKey::from_uncompressed_bytes`).
    pub fn append(&mut self, keys: &[&PublicKey]) -> u64 {
        let flat: Vec<u8> = keys.iter().flat_map(|k| k.point.iter().copied()).collect();
        let mut first = 0u64;
        let mut errs = vec![0u8; keys.len()];
        let rc = unsafe { mbls_keytable_append(self.h, flat.as_ptr(), PK_UNCOMPRESSED, 0, keys.len() as u64, &mut first, errs.as_mut_ptr()) };
        if rc != 0 {
            err(rc);
        }
        first
    }
    pub fn get(&self, index: u64) -> Result<PublicKey, AmclError> {
        let mut p = [0u8; 96];
        let mut e = 0u8;
        let rc = unsafe { mbls_keytable_get(self.h, index, 1, p.as_mut_ptr(), &mut e) };
        if rc != 0 {
            err(rc);
        }
        check(e as c_int)?;
        Ok(PublicKey { point: p })
    }
    /// n x fast_aggregate_verify with `k` table indices per item.
    pub fn fast_aggregate_verify(&self, signatures: &[AggregateSignature], messages: &[[u8; 32]], key_indices: &[u32], k: u32) -> Vec<bool> {
        let n = signatures.len();
        assert!(messages.len() == n && key_indices.len() == n * k as usize);
        let sigs: Vec<u8> = signatures.iter().flat_map(|s| s.point.iter().copied()).collect();
        let msgs: Vec<u8> = messages.iter().flat_map(|m| m.iter().copied()).collect();
        let mut res = vec![0u8; n];
        let rc = unsafe {
            mbls_fast_aggregate_verify_batch_indexed(ctx(), self.h, sigs.as_ptr(), msgs.as_ptr(), 32, std::ptr::null(), key_indices.as_ptr(), std::ptr::null(), n as u64, k, res.as_mut_ptr(), std::ptr::null_mut())
        };
        if rc != 0 {
            err(rc);
        }
        res.into_iter().map(|b| b == 1).collect()
    }
}
impl Drop for KeyTable {
    fn drop(&mut self) {
        unsafe { mbls_keytable_destroy(self.h) }
    }
}

#[allow(dead_code)]
fn _unused(_: *mut c_void) {}

#[cfg(test)]
mod tests {
    //! The reference's own unit tests that need no fixture files, restated (they need an MI355X to run).
    use super::*;

    #[test]
    fn test_readme_example() {
        // reference src/signature.rs:103-125
        let sk_bytes = [78, 252, 122, 126, 32, 0, 75, 89, 252, 31, 42, 130, 254, 88, 6, 90, 138, 202, 135, 194, 233, 117, 181, 75, 96, 238, 79, 100, 237, 59, 140, 111];
        let sk = SecretKey::from_bytes(&sk_bytes).unwrap();
        let pk = PublicKey::from_secret_key(&sk);
        let msg = "cats".as_bytes();
        let sig = Signature::new(msg, &sk);
        assert!(sig.verify(msg, &pk));
        assert!(!sig.verify("dogs".as_bytes(), &pk));
    }

    #[test]
    fn test_fast_aggregate_verify_infinity_aggregate_key() {
        // reference src/aggregates.rs:392-410: pk(1) + pk(r-1) = infinity -> false
        let mut one = [0u8; 32];
        one[31] = 1;
        let mut rm1 = CURVE_ORDER_BE;
        rm1[31] = 0;
        let sk1 = SecretKey::from_bytes(&one).unwrap();
        let sk2 = SecretKey::from_bytes(&rm1).unwrap();
        let (pk1, pk2) = (PublicKey::from_secret_key(&sk1), PublicKey::from_secret_key(&sk2));
        let msg = [1u8; 32];
        let mut agg = AggregateSignature::new();
        agg.add(&Signature::new(&msg, &sk1));
        agg.add(&Signature::new(&msg, &sk2));
        assert!(!agg.fast_aggregate_verify(&msg, &[&pk1, &pk2]));
        assert!(!AggregateSignature::new().fast_aggregate_verify(&msg, &[]));
    }
}
