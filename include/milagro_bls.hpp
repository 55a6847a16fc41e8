// milagro_bls.hpp -- C++ host-side mirror of the reference's public API (reference src/lib.rs:17-22) over the C ABI of
// libmbls_hip.so (include/mbls.h). Header-only; same type and method names, argument meaning and error behaviour as
// the Rust crate, so code written against milagro_bls reads the same:
//
//     reference (Rust)                                   here (C++17)
//     PublicKey::from_bytes(&[u8]) -> Result<..>         PublicKey::from_bytes(bytes)  (throws AmclError)
//     Signature::new(msg, &sk)                           Signature::new_(msg, sk)      ("new" is a keyword)
//     sig.verify(msg, &pk) -> bool                       sig.verify(msg, pk) -> bool
//     AggregateSignature::fast_aggregate_verify(..)      same
//     AggregateSignature::verify_multiple_aggregate_signatures(rng, iter)   same, rng = any callable returning uint8_t
//
// Every curve / pairing operation runs in the HIP kernels; there is no CPU fallback: constructing the first object without a
// GPU throws DeviceError. The only host arithmetic is SecretKey::key_generate (HKDF-SHA-256 + one reduction mod r), which the
// reference also does on the host (src/keys.rs:45-77).
// Threads: all objects go through one process-wide mbls_ctx; every C ABI entry takes that context's lock, so the types can be used
// from any number of threads like the reference's -- calls are serialised on the one GPU.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <exception>
#include <random>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>
#include "mbls.h"

namespace milagro_bls {

constexpr size_t G1_BYTES = MBLS_G1_BYTES;                 // reference src/lib.rs:20
constexpr size_t G2_BYTES = MBLS_G2_BYTES;
constexpr size_t SECRET_KEY_BYTES = MBLS_SECRET_KEY_BYTES;
using Bytes = std::vector<uint8_t>;

// amcl::errors::AmclError (reference src/amcl_utils.rs:11): the variants the reference uses
struct AmclError : std::runtime_error {
    enum Kind { InvalidG1Size = 1, InvalidG2Size = 2, InvalidPoint = 3, AggregateEmptyPoints = 4, InvalidSecretKeySize = 5, InvalidSecretKeyRange = 6 };
    Kind kind;
    explicit AmclError(int k) : std::runtime_error("AmclError " + std::to_string(k)), kind(static_cast<Kind>(k)) {}
};
struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };

namespace detail {
inline mbls_ctx* ctx() {
    static mbls_ctx* c = [] {
        mbls_ctx* p = nullptr;
        if (mbls_ctx_create(&p, 0) != MBLS_OK) throw DeviceError("mbls_ctx_create failed: no MI355X / HIP device (there is no CPU fallback)");
        return p;
    }();
    return c;
}
inline void check(int rc) {
    if (rc == MBLS_OK) return;
    if (rc >= MBLS_ERR_DEVICE) throw DeviceError(std::string("mbls device error: ") + mbls_last_error(ctx()));
    throw AmclError(rc);
}
}  // namespace detail

// ---- host-side SHA-256 / HMAC / HKDF for KeyGenerate (reference src/keys.rs:45-77 uses amcl's HASH256::hkdf_extract / hkdf_extend)
namespace detail {
struct Sha256 {
    uint32_t h[8]; uint8_t buf[64]; uint64_t len = 0; size_t fill = 0;
    Sha256() { static const uint32_t iv[8] = {0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19}; std::memcpy(h, iv, 32); }
    static uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
            0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
            0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
            0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = (uint32_t(p[4 * i]) << 24) | (uint32_t(p[4 * i + 1]) << 16) | (uint32_t(p[4 * i + 2]) << 8) | p[4 * i + 3];
        for (int i = 16; i < 64; i++) w[i] = w[i - 16] + (ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10));
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const uint8_t* p, size_t n) {
        len += n;
        while (n) { size_t t = std::min(n, size_t(64) - fill); std::memcpy(buf + fill, p, t); fill += t; p += t; n -= t; if (fill == 64) { block(buf); fill = 0; } }
    }
    std::array<uint8_t, 32> finish() {
        uint64_t bits = len * 8; uint8_t pad = 0x80; update(&pad, 1);
        uint8_t z = 0; while (fill != 56) update(&z, 1);
        uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = uint8_t(bits >> (56 - 8 * i)); update(lb, 8);
        std::array<uint8_t, 32> o; for (int i = 0; i < 8; i++) { o[4 * i] = uint8_t(h[i] >> 24); o[4 * i + 1] = uint8_t(h[i] >> 16); o[4 * i + 2] = uint8_t(h[i] >> 8); o[4 * i + 3] = uint8_t(h[i]); }
        return o;
    }
};
inline std::array<uint8_t, 32> sha256(const Bytes& m) { Sha256 s; s.update(m.data(), m.size()); return s.finish(); }
inline std::array<uint8_t, 32> hmac_sha256(const Bytes& key, const Bytes& msg) {
    Bytes k = key; if (k.size() > 64) { auto d = sha256(k); k.assign(d.begin(), d.end()); } k.resize(64, 0);
    Bytes i(64), o(64); for (int j = 0; j < 64; j++) { i[j] = k[j] ^ 0x36; o[j] = k[j] ^ 0x5c; }
    i.insert(i.end(), msg.begin(), msg.end()); auto inner = sha256(i);
    o.insert(o.end(), inner.begin(), inner.end()); return sha256(o);
}
// HKDF-SHA-256 (RFC 5869; amcl's HASH256::hkdf_extract / hkdf_extend, reference src/keys.rs:62-69). An empty salt is HashLen zero bytes.
inline std::array<uint8_t, 32> hkdf_extract(const Bytes& salt, const Bytes& ikm) { return hmac_sha256(salt.empty() ? Bytes(32, 0) : salt, ikm); }
inline Bytes hkdf_expand(const std::array<uint8_t, 32>& prk, const Bytes& info, size_t len) {
    Bytes okm, t; uint8_t ctr = 1;
    while (okm.size() < len) {
        Bytes m = t; m.insert(m.end(), info.begin(), info.end()); m.push_back(ctr++);
        auto d = hmac_sha256(Bytes(prk.begin(), prk.end()), m); t.assign(d.begin(), d.end());
        okm.insert(okm.end(), t.begin(), t.end());
    }
    okm.resize(len); return okm;
}
// OS2IP(48 bytes) mod r -> 32 bytes big-endian (bitwise long division on 32-bit limbs; host-only, once per key)
inline std::array<uint8_t, 32> mod_r(const uint8_t okm[48]) {
    static const uint32_t R[9] = {0x00000001, 0xffffffff, 0xfffe5bfe, 0x53bda402, 0x09a1d805, 0x3339d808, 0x299d7d48, 0x73eda753, 0};
    uint32_t a[9] = {0};
    for (int bit = 0; bit < 384; bit++) {
        for (int j = 8; j > 0; j--) a[j] = (a[j] << 1) | (a[j - 1] >> 31);
        a[0] = (a[0] << 1) | ((okm[bit >> 3] >> (7 - (bit & 7))) & 1u);
        bool ge = true; for (int j = 8; j >= 0; j--) { if (a[j] != R[j]) { ge = a[j] > R[j]; break; } }
        if (ge) { uint64_t br = 0; for (int j = 0; j < 9; j++) { uint64_t d = uint64_t(a[j]) - R[j] - br; a[j] = uint32_t(d); br = (d >> 63) & 1; } }
    }
    std::array<uint8_t, 32> o; for (int j = 0; j < 8; j++) { uint32_t v = a[7 - j]; o[4 * j] = uint8_t(v >> 24); o[4 * j + 1] = uint8_t(v >> 16); o[4 * j + 2] = uint8_t(v >> 8); o[4 * j + 3] = uint8_t(v); }
    return o;
}
}  // namespace detail

// reference src/keys.rs:28-113 (host-only object; the scalar goes to the GPU only for signing)
class SecretKey {
    std::array<uint8_t, 32> x_{};
public:
    // KeyGenerate, reference src/keys.rs:45-77: HKDF-SHA-256 with the salt-rehash loop (KEY_SALT :24, L = 48 :26)
    static SecretKey key_generate(const Bytes& ikm, const Bytes& key_info = {}) {
        if (ikm.size() < 32) throw AmclError(AmclError::InvalidSecretKeySize);
        const char* ks = "BLS-SIG-KEYGEN-SALT-";
        Bytes salt(ks, ks + 20);
        SecretKey s;
        bool zero = true;
        while (zero) {
            auto hs = detail::sha256(salt); salt.assign(hs.begin(), hs.end());                          // salt = H(salt)
            Bytes ikm0 = ikm; ikm0.push_back(0);
            auto prk = detail::hkdf_extract(salt, ikm0);                                                // PRK = HKDF-Extract(salt, IKM || 0)
            Bytes info = key_info; info.push_back(0); info.push_back(48);                               // key_info || I2OSP(L, 2)
            Bytes okm = detail::hkdf_expand(prk, info, 48);                                             // OKM = HKDF-Expand(PRK, info, L)
            s.x_ = detail::mod_r(okm.data());                                                           // SK = OS2IP(OKM) mod r
            for (uint8_t v : s.x_) if (v) zero = false;
        }
        return s;
    }
    // SecretKey::random (src/keys.rs:36-39): 32 bytes of IKM from the caller's byte source (rng() returns one random byte)
    template <typename Rng> static SecretKey random(Rng&& rng) { Bytes ikm(32); for (auto& b : ikm) b = uint8_t(rng()); return key_generate(ikm); }
    static SecretKey random() { std::random_device rd; return random([&] { return uint8_t(rd()); }); }
    static SecretKey from_bytes(const Bytes& b) {                          // src/keys.rs:80-82, error cases :285-297
        static const uint8_t R[32] = {0x73,0xed,0xa7,0x53,0x29,0x9d,0x7d,0x48,0x33,0x39,0xd8,0x08,0x09,0xa1,0xd8,0x05,0x53,0xbd,0xa4,0x02,0xff,0xfe,0x5b,0xfe,0xff,0xff,0xff,0xff,0x00,0x00,0x00,0x01};
        if (b.size() != 32) throw AmclError(AmclError::InvalidSecretKeySize);
        bool zero = true; for (uint8_t v : b) if (v) zero = false;
        if (zero || std::memcmp(b.data(), R, 32) >= 0) throw AmclError(AmclError::InvalidSecretKeyRange);
        SecretKey s; std::memcpy(s.x_.data(), b.data(), 32); return s;
    }
    Bytes as_bytes() const { return Bytes(x_.begin(), x_.end()); }
    bool operator==(const SecretKey& o) const { return x_ == o.x_; }
    ~SecretKey() { volatile uint8_t* p = x_.data(); for (int i = 0; i < 32; i++) p[i] = 0; }   // zeroize on drop, src/keys.rs:109-113
};

// reference src/keys.rs:116-187; `point` is the 96-byte uncompressed form
struct PublicKey {
    std::array<uint8_t, 96> point{};
    static PublicKey from_secret_key(const SecretKey& sk) { PublicKey p; Bytes b = sk.as_bytes(); detail::check(mbls_pk_from_secret_key(detail::ctx(), b.data(), b.size(), p.point.data())); return p; }
    static PublicKey from_bytes(const Bytes& b) { PublicKey p; detail::check(mbls_pk_from_bytes(detail::ctx(), b.data(), b.size(), p.point.data())); return p; }
    static PublicKey from_bytes_unchecked(const Bytes& b) { PublicKey p; detail::check(mbls_pk_from_bytes_unchecked(detail::ctx(), b.data(), b.size(), p.point.data())); return p; }
    static PublicKey from_uncompressed_bytes(const Bytes& b) { PublicKey p; detail::check(mbls_pk_from_uncompressed_bytes(detail::ctx(), b.data(), b.size(), p.point.data())); return p; }
    std::array<uint8_t, 48> as_bytes() const { std::array<uint8_t, 48> o{}; detail::check(mbls_pk_as_bytes(detail::ctx(), point.data(), o.data())); return o; }
    std::array<uint8_t, 96> as_uncompressed_bytes() const { return point; }
    bool key_validate() const { return mbls_pk_key_validate(detail::ctx(), point.data()) == 1; }
    bool operator==(const PublicKey& o) const { return point == o.point; }
};

struct Keypair {                                                             // reference src/keys.rs:189-204
    SecretKey sk; PublicKey pk;
    template <typename Rng> static Keypair random(Rng&& rng) { SecretKey s = SecretKey::random(rng); PublicKey p = PublicKey::from_secret_key(s); return Keypair{s, p}; }
    static Keypair random() { SecretKey s = SecretKey::random(); PublicKey p = PublicKey::from_secret_key(s); return Keypair{s, p}; }
};

// reference src/signature.rs:9-51; `point` is the 96-byte compressed form
struct Signature {
    std::array<uint8_t, 96> point{};
    static Signature new_(const Bytes& msg, const SecretKey& sk) {
        Signature s; Bytes k = sk.as_bytes(); detail::check(mbls_sign(detail::ctx(), msg.data(), msg.size(), k.data(), k.size(), s.point.data())); return s;
    }
    bool verify(const Bytes& msg, const PublicKey& pk) const { return mbls_verify(detail::ctx(), point.data(), msg.data(), msg.size(), pk.point.data()) == 1; }
    static Signature from_bytes(const Bytes& b) { Signature s; detail::check(mbls_sig_from_bytes(detail::ctx(), b.data(), b.size(), s.point.data())); return s; }
    std::array<uint8_t, 96> as_bytes() const { return point; }
    bool operator==(const Signature& o) const { return point == o.point; }
};

// reference src/aggregates.rs:17-78
struct AggregatePublicKey {
    std::array<uint8_t, 96> point{};
    static AggregatePublicKey aggregate(const std::vector<const PublicKey*>& keys) {
        if (keys.empty()) throw AmclError(AmclError::AggregateEmptyPoints);
        Bytes flat; for (auto* k : keys) flat.insert(flat.end(), k->point.begin(), k->point.end());
        AggregatePublicKey a; detail::check(mbls_aggregate_public_keys(detail::ctx(), flat.data(), keys.size(), a.point.data())); return a;
    }
    static AggregatePublicKey into_aggregate(const std::vector<PublicKey>& keys) {
        std::vector<const PublicKey*> p; for (auto& k : keys) p.push_back(&k); return aggregate(p);
    }
    static AggregatePublicKey from_public_key(const PublicKey& k) { AggregatePublicKey a; a.point = k.point; return a; }
    void add(const PublicKey& k) { detail::check(mbls_aggregate_public_key_add(detail::ctx(), point.data(), k.point.data(), point.data())); }
    void add_aggregate(const AggregatePublicKey& o) { detail::check(mbls_aggregate_public_key_add(detail::ctx(), point.data(), o.point.data(), point.data())); }
    bool operator==(const AggregatePublicKey& o) const { return point == o.point; }
};

// reference src/aggregates.rs:83-334
struct AggregateSignature {
    std::array<uint8_t, 96> point{};
    AggregateSignature() { point[0] = 0xC0; }                                // AggregateSignature::new(): the point at infinity
    static AggregateSignature aggregate(const std::vector<const Signature*>& sigs) {      // src/aggregates.rs:100-106: one batched launch
        AggregateSignature a; if (sigs.empty()) return a;
        Bytes flat; for (auto* s : sigs) flat.insert(flat.end(), s->point.begin(), s->point.end());
        uint8_t e = 0; detail::check(mbls_aggregate_signatures_batch(detail::ctx(), flat.data(), nullptr, 1, uint32_t(sigs.size()), a.point.data(), &e)); detail::check(e);
        return a;
    }
    static AggregateSignature from_signature(const Signature& s) { AggregateSignature a; a.point = s.point; return a; }
    void add(const Signature& s) { detail::check(mbls_aggregate_signature_add(detail::ctx(), point.data(), s.point.data(), point.data())); }
    void add_aggregate(const AggregateSignature& o) { detail::check(mbls_aggregate_signature_add(detail::ctx(), point.data(), o.point.data(), point.data())); }
    bool aggregate_verify(const std::vector<Bytes>& msgs, const std::vector<const PublicKey*>& pks) const {
        Bytes flat_m, flat_p; std::vector<size_t> lens;
        for (auto& m : msgs) { flat_m.insert(flat_m.end(), m.begin(), m.end()); lens.push_back(m.size()); }
        for (auto* k : pks) flat_p.insert(flat_p.end(), k->point.begin(), k->point.end());
        return mbls_aggregate_verify(detail::ctx(), point.data(), flat_m.data(), lens.data(), msgs.size(), flat_p.data(), pks.size()) == 1;
    }
    bool fast_aggregate_verify(const Bytes& msg, const std::vector<const PublicKey*>& pks) const {
        Bytes flat; for (auto* k : pks) flat.insert(flat.end(), k->point.begin(), k->point.end());
        return mbls_fast_aggregate_verify(detail::ctx(), point.data(), msg.data(), msg.size(), flat.data(), pks.size()) == 1;
    }
    bool fast_aggregate_verify_pre_aggregated(const Bytes& msg, const AggregatePublicKey& apk) const {
        return mbls_fast_aggregate_verify_pre_aggregated(detail::ctx(), point.data(), msg.data(), msg.size(), apk.point.data()) == 1;
    }
    // one scalar as at reference src/aggregates.rs:280-287: 8 random bytes, big-endian i64, absolute value (as the release build wraps it), again on zero
    template <typename Rng> static uint64_t draw_scalar(Rng& rng) {
        uint64_t r = 0;
        while (r == 0) { uint64_t v = 0; for (int j = 0; j < 8; j++) v = (v << 8) | uint8_t(rng()); r = (v >> 63) ? (uint64_t(0) - v) : v; }
        return r;
    }
    // One call (mbls_verify_multiple_aggregate_signatures_rng): the library tests the signatures first and asks for the scalars of the sets in front of the first
    // bad one only -- the reference's order (its loop tests set i's signature for the subgroup, :272-275, BEFORE it draws rand[i] and returns at the first signature
    // outside G2: a rejected batch leaves the caller's generator where the reference would) without a second subgroup test. rng(): one random byte per call.
    template <typename Rng>
    static bool verify_multiple_aggregate_signatures(Rng&& rng, const std::vector<std::tuple<const AggregateSignature*, const AggregatePublicKey*, Bytes>>& sets) {
        if (sets.empty()) return mbls_verify_multiple_aggregate_signatures(detail::ctx(), nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0) == 1;
        Bytes sigs, apks, msgs; std::vector<uint64_t> moff{0};
        for (auto& s : sets) {
            sigs.insert(sigs.end(), std::get<0>(s)->point.begin(), std::get<0>(s)->point.end());
            apks.insert(apks.end(), std::get<1>(s)->point.begin(), std::get<1>(s)->point.end());
            msgs.insert(msgs.end(), std::get<2>(s).begin(), std::get<2>(s).end());
            moff.push_back(msgs.size());
        }
        using R = typename std::remove_reference<Rng>::type;
        struct src { R* rng; std::exception_ptr err; } u{&rng, nullptr};
        mbls_scalar_source draw = [](void* user, uint64_t* out, uint64_t count) {
            src* p = static_cast<src*>(user);
            try { for (uint64_t i = 0; i < count; i++) out[i] = draw_scalar(*p->rng); }
            catch (...) { p->err = std::current_exception(); for (uint64_t i = 0; i < count; i++) out[i] = 0; }        // never unwind through the C frames
        };
        const bool ok = mbls_verify_multiple_aggregate_signatures_rng(detail::ctx(), sigs.data(), apks.data(), msgs.data(), 0, moff.data(), sets.size(), draw, &u) == 1;
        if (u.err) std::rethrow_exception(u.err);
        return ok;
    }
    // the same check with the sets cut into one shard per device of a multi-device handle (mbls_multi_create): same bool, same RNG order, ONE call
    // (mbls_multi_verify_multiple_aggregate_signatures_rng: every device tests its shard's signatures first, the scalars are asked for once)
    template <typename Rng>
    static bool verify_multiple_aggregate_signatures(mbls_multi* devices, Rng&& rng, const std::vector<std::tuple<const AggregateSignature*, const AggregatePublicKey*, Bytes>>& sets) {
        if (sets.empty()) return true;
        Bytes sigs, apks, msgs; std::vector<uint64_t> moff{0};
        for (auto& s : sets) {
            sigs.insert(sigs.end(), std::get<0>(s)->point.begin(), std::get<0>(s)->point.end());
            apks.insert(apks.end(), std::get<1>(s)->point.begin(), std::get<1>(s)->point.end());
            msgs.insert(msgs.end(), std::get<2>(s).begin(), std::get<2>(s).end());
            moff.push_back(msgs.size());
        }
        using R = typename std::remove_reference<Rng>::type;
        struct src { R* rng; std::exception_ptr err; } u{&rng, nullptr};
        mbls_scalar_source draw = [](void* user, uint64_t* out, uint64_t count) {
            src* p = static_cast<src*>(user);
            try { for (uint64_t i = 0; i < count; i++) out[i] = draw_scalar(*p->rng); }
            catch (...) { p->err = std::current_exception(); for (uint64_t i = 0; i < count; i++) out[i] = 0; }
        };
        const bool ok = mbls_multi_verify_multiple_aggregate_signatures_rng(devices, sigs.data(), apks.data(), msgs.data(), 0, moff.data(), sets.size(), draw, &u) == 1;
        if (u.err) std::rethrow_exception(u.err);
        return ok;
    }
    static AggregateSignature from_bytes(const Bytes& b) { AggregateSignature a; detail::check(mbls_sig_from_bytes(detail::ctx(), b.data(), b.size(), a.point.data())); return a; }
    std::array<uint8_t, 96> as_bytes() const { return point; }
    bool operator==(const AggregateSignature& o) const { return point == o.point; }
};

// Several GPUs behind one handle (mbls_multi_*; not part of the reference's API): contiguous shards, one context and one host thread per
// device inside the library, results written in place. Items are independent (reference src/aggregates.rs:177-215 keeps no state).
// n x AggregateSignature::aggregate_verify (src/aggregates.rs:130-170) in one launch chain: item i = (sigs[i], its messages, its keys); an item
// with a different number of messages and keys, or with none, is false like the reference's early return (:131-133)
inline std::vector<bool> aggregate_verify_batch(const std::vector<AggregateSignature>& sigs, const std::vector<std::vector<Bytes>>& msgs,
                                                const std::vector<std::vector<const PublicKey*>>& keys) {
    const size_t n = sigs.size();
    if (msgs.size() != n || keys.size() != n) throw std::invalid_argument("one message list and one key list per signature");
    Bytes s, m, p; std::vector<uint64_t> moff{0}; std::vector<uint32_t> poff{0}; std::vector<bool> mismatch(n, false);
    for (size_t i = 0; i < n; i++) {
        s.insert(s.end(), sigs[i].point.begin(), sigs[i].point.end());
        if (msgs[i].size() != keys[i].size()) mismatch[i] = true;          // enters the batch without pairs: false
        else for (size_t j = 0; j < keys[i].size(); j++) {
            m.insert(m.end(), msgs[i][j].begin(), msgs[i][j].end()); moff.push_back(m.size());
            p.insert(p.end(), keys[i][j]->point.begin(), keys[i][j]->point.end());
        }
        if (p.size() / 96 > 0xFFFFFFFFull) throw std::invalid_argument("aggregate_verify_batch: pair indices are 32-bit");
        poff.push_back(uint32_t(p.size() / 96));
    }
    std::vector<uint8_t> res(n ? n : 1);
    detail::check(mbls_aggregate_verify_batch(detail::ctx(), s.data(), m.data(), 0, moff.data(), p.data(), poff.data(), 0, n, res.data(), nullptr));
    std::vector<bool> out(n);
    for (size_t i = 0; i < n; i++) out[i] = res[i] == 1 && !mismatch[i];
    return out;
}

class MultiGpu {
    mbls_multi* h_ = nullptr;
public:
    explicit MultiGpu(const std::vector<int>& device_ids) {
        if (mbls_multi_create(&h_, device_ids.data(), int(device_ids.size())) != MBLS_OK) throw DeviceError("mbls_multi_create failed");
    }
    MultiGpu(const MultiGpu&) = delete; MultiGpu& operator=(const MultiGpu&) = delete;
    ~MultiGpu() { mbls_multi_destroy(h_); }
    int devices() const { return mbls_multi_device_count(h_); }
    // the handle's exchange steps as RCCL all-gathers between the devices (true) or through host memory (false; exchange_note() says why)
    bool rccl_active() const { return mbls_multi_rccl_active(h_) == 1; }
    std::string exchange_note() const { return mbls_multi_exchange_note(h_); }
    mbls_multi* handle() const { return h_; }
    // n x fast_aggregate_verify with the results as one packed accept bitmap that every device ends up holding (bit i % 64 of word i / 64 = item i)
    std::vector<uint64_t> fast_aggregate_verify_bitmap(const std::vector<AggregateSignature>& sigs, const std::vector<Bytes>& msgs, const std::vector<std::vector<const PublicKey*>>& keys) const {
        const size_t n = sigs.size();
        if (msgs.size() != n || keys.size() != n) throw std::invalid_argument("one message and one key set per signature");
        Bytes s, m, p; std::vector<uint64_t> moff{0}; std::vector<uint32_t> koff{0};
        for (size_t i = 0; i < n; i++) {
            s.insert(s.end(), sigs[i].point.begin(), sigs[i].point.end());
            m.insert(m.end(), msgs[i].begin(), msgs[i].end()); moff.push_back(m.size());
            for (auto* k : keys[i]) p.insert(p.end(), k->point.begin(), k->point.end());
            if (p.size() / 96 > 0xFFFFFFFFull) throw std::invalid_argument("fast_aggregate_verify_bitmap: key indices are 32-bit");
            koff.push_back(uint32_t(p.size() / 96));
        }
        std::vector<uint64_t> words((n + 63) / 64 ? (n + 63) / 64 : 1);
        int rc = mbls_multi_fast_aggregate_verify_bitmap(h_, s.data(), m.data(), 0, moff.data(), p.data(), MBLS_PK_UNCOMPRESSED, koff.data(), n, 0, words.data(), nullptr);
        if (rc != MBLS_OK) throw DeviceError(std::string("mbls_multi: ") + mbls_multi_last_error(h_));
        words.resize((n + 63) / 64);
        return words;
    }
    // n x AggregateSignature::fast_aggregate_verify; messages of any length each
    std::vector<bool> fast_aggregate_verify(const std::vector<AggregateSignature>& sigs, const std::vector<Bytes>& msgs, const std::vector<std::vector<const PublicKey*>>& keys) const {
        const size_t n = sigs.size();
        if (msgs.size() != n || keys.size() != n) throw std::invalid_argument("one message and one key set per signature");
        Bytes s, m, p; std::vector<uint64_t> moff{0}; std::vector<uint32_t> koff{0};
        for (size_t i = 0; i < n; i++) {
            s.insert(s.end(), sigs[i].point.begin(), sigs[i].point.end());
            m.insert(m.end(), msgs[i].begin(), msgs[i].end()); moff.push_back(m.size());
            for (auto* k : keys[i]) p.insert(p.end(), k->point.begin(), k->point.end());
            if (p.size() / 96 > 0xFFFFFFFFull) throw std::invalid_argument("fast_aggregate_verify: key indices are 32-bit");
            koff.push_back(uint32_t(p.size() / 96));
        }
        std::vector<uint8_t> res(n ? n : 1);
        int rc = mbls_multi_fast_aggregate_verify_batch(h_, s.data(), m.data(), 0, moff.data(), p.data(), MBLS_PK_UNCOMPRESSED, koff.data(), n, 0, res.data(), nullptr);
        if (rc != MBLS_OK) throw DeviceError(std::string("mbls_multi: ") + mbls_multi_last_error(h_));
        return std::vector<bool>(res.begin(), res.begin() + n);
    }
};

}  // namespace milagro_bls
