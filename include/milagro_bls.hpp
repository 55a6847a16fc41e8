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
// Every numeric operation runs in the HIP kernels; there is no CPU arithmetic here and no fallback: constructing the
// first object without a GPU throws DeviceError.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <tuple>
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

// reference src/keys.rs:28-113 (host-only; HKDF key generation is out of this library's scope)
class SecretKey {
    std::array<uint8_t, 32> x_{};
public:
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

struct Keypair { SecretKey sk; PublicKey pk; };                              // reference src/keys.rs:189-204

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
    static AggregateSignature aggregate(const std::vector<const Signature*>& sigs) { AggregateSignature a; for (auto* s : sigs) a.add(*s); return a; }
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
    // rng(): one random byte per call. Blinding scalars drawn as at reference src/aggregates.rs:280-287.
    template <typename Rng>
    static bool verify_multiple_aggregate_signatures(Rng&& rng, const std::vector<std::tuple<const AggregateSignature*, const AggregatePublicKey*, Bytes>>& sets) {
        if (sets.empty()) return mbls_verify_multiple_aggregate_signatures(detail::ctx(), nullptr, nullptr, nullptr, 0, nullptr, 0) == 1;
        Bytes sigs, apks, msgs; std::vector<uint64_t> rands;
        const size_t mlen = std::get<2>(sets[0]).size();
        for (auto& s : sets) {
            uint64_t r = 0;
            while (r == 0) { uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | uint8_t(rng()); int64_t sv = int64_t(v); r = uint64_t(sv < 0 ? -sv : sv); }
            rands.push_back(r);
            sigs.insert(sigs.end(), std::get<0>(s)->point.begin(), std::get<0>(s)->point.end());
            apks.insert(apks.end(), std::get<1>(s)->point.begin(), std::get<1>(s)->point.end());
            if (std::get<2>(s).size() != mlen) throw std::invalid_argument("messages must have equal length");
            msgs.insert(msgs.end(), std::get<2>(s).begin(), std::get<2>(s).end());
        }
        return mbls_verify_multiple_aggregate_signatures(detail::ctx(), sigs.data(), apks.data(), msgs.data(), uint32_t(mlen), rands.data(), sets.size()) == 1;
    }
    static AggregateSignature from_bytes(const Bytes& b) { AggregateSignature a; detail::check(mbls_sig_from_bytes(detail::ctx(), b.data(), b.size(), a.point.data())); return a; }
    std::array<uint8_t, 96> as_bytes() const { return point; }
    bool operator==(const AggregateSignature& o) const { return point == o.point; }
};

}  // namespace milagro_bls
