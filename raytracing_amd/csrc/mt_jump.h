// Host side of MT19937 jump-ahead: the polynomials x^(2^k) mod phi(x), phi the
// characteristic polynomial (degree 19937) of the generator's one-word state step.
//
// The word sequence x[n] of MT19937 obeys a linear recurrence over GF(2).  With W_n the
// window (x[n], ..., x[n+623]) and f the map W_n -> W_{n+1}, every polynomial g with
// g(x) = x^J mod phi(x) satisfies g(f) W_n = W_{n+J} except for the low 31 bits of the
// window's first word (the part of the window the recurrence never reads: f's minimal
// polynomial is x * phi(x)).  g(f) W = XOR over the set coefficients g_i of f^i W, i.e. a
// GF(2) convolution of the sequence generated from W with the coefficient bits - that part
// runs on the device (k_mt_jump); here are phi (Berlekamp-Massey on one output bit) and the
// table of x^(2^k) mod phi (repeated squaring).  Method: Haramoto, Matsumoto, Nishimura,
// Panneton, L'Ecuyer, "Efficient jump ahead for F2-linear random number generators" (2008).
#pragma once
#include <stdint.h>
#include <string.h>

#include <array>
#include <mutex>
#include <vector>

namespace mtjump {

constexpr int DEG = 19937;
constexpr int PW = 312;                      // 64-bit words holding 19968 >= DEG + 1 bits
using Poly = std::array<uint64_t, PW>;       // bit i = coefficient of x^i
constexpr int MAX_K = 48;                    // table covers jumps up to 2^47 words

inline bool get_bit(const uint64_t* p, int i) { return (p[i >> 6] >> (i & 63)) & 1u; }
inline void flip_bit(uint64_t* p, int i) { p[i >> 6] ^= (uint64_t)1 << (i & 63); }

// dst ^= src << shift   (src has n_src words, dst has n_dst words)
inline void xor_shifted(uint64_t* dst, int n_dst, const uint64_t* src, int n_src, int shift) {
    const int ws = shift >> 6, bs = shift & 63;
    for (int i = 0; i < n_src; ++i) {
        const uint64_t v = src[i];
        if (!v) continue;
        if (i + ws < n_dst) dst[i + ws] ^= v << bs;
        if (bs && i + ws + 1 < n_dst) dst[i + ws + 1] ^= v >> (64 - bs);
    }
}

// Plain MT19937 word sequence (untempered) from the standard integer seeding.
struct HostSequence {
    uint32_t x[624];
    int i = 624;
    explicit HostSequence(uint32_t seed) {
        x[0] = seed;
        for (int j = 1; j < 624; ++j) x[j] = 1812433253u * (x[j - 1] ^ (x[j - 1] >> 30)) + (uint32_t)j;
    }
    uint32_t next() {
        if (i == 624) {
            for (int k = 0; k < 624; ++k) {
                const uint32_t y = (x[k] & 0x80000000u) | (x[(k + 1) % 624] & 0x7fffffffu);
                x[k] = x[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            i = 0;
        }
        return x[i++];
    }
};

// Minimal polynomial of the bit sequence s[0..n): returns its degree L, coefficients in c
// (connection polynomial: s[j] = XOR_{i=1..L} c_i s[j-i]).
inline int berlekamp_massey(const std::vector<uint8_t>& s, Poly& c_out) {
    const int n = (int)s.size();
    Poly C{}, B{}, T{};
    std::array<uint64_t, PW> rev{};   // bit j = s[pos - 1 - j]
    C[0] = 1; B[0] = 1;
    int L = 0, m = 1;
    for (int pos = 0; pos < n; ++pos) {
        // discrepancy d = s[pos] ^ XOR_{i=1..L} c_i s[pos-i]  = s[pos] ^ parity((C >> 1) & rev)
        uint64_t acc = 0;
        for (int w = 0; w < PW; ++w) {
            const uint64_t cs = (C[w] >> 1) | (w + 1 < PW ? C[w + 1] << 63 : 0);
            acc ^= cs & rev[w];
        }
        const int d = (int)(s[pos] ^ (__builtin_popcountll(acc) & 1));
        if (d) {
            if (2 * L <= pos) {
                T = C;
                xor_shifted(C.data(), PW, B.data(), PW, m);
                L = pos + 1 - L;
                B = T;
                m = 1;
            } else {
                xor_shifted(C.data(), PW, B.data(), PW, m);
                ++m;
            }
        } else {
            ++m;
        }
        // rev <<= 1, insert s[pos]
        for (int w = PW - 1; w > 0; --w) rev[w] = (rev[w] << 1) | (rev[w - 1] >> 63);
        rev[0] = (rev[0] << 1) | s[pos];
    }
    c_out = C;
    return L;
}

struct Tables {
    bool ok = false;
    Poly phi{};                       // degree 19937
    std::vector<Poly> pow2;           // pow2[k] = x^(2^k) mod phi
};

// a(x)^2 mod phi
inline Poly square_mod(const Poly& a, const Poly& phi) {
    std::array<uint64_t, 2 * PW> t{};
    for (int w = 0; w < PW; ++w) {
        // spread the bits of a[w] into two words (bit i -> bit 2i)
        uint64_t lo = a[w] & 0xffffffffu, hi = a[w] >> 32;
        auto spread = [](uint64_t v) {
            v = (v | (v << 16)) & 0x0000ffff0000ffffull;
            v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
            v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
            v = (v | (v << 2)) & 0x3333333333333333ull;
            v = (v | (v << 1)) & 0x5555555555555555ull;
            return v;
        };
        t[2 * w] = spread(lo);
        t[2 * w + 1] = spread(hi);
    }
    for (int i = 2 * DEG - 2; i >= DEG; --i)
        if (get_bit(t.data(), i)) xor_shifted(t.data(), 2 * PW, phi.data(), PW, i - DEG);
    Poly r{};
    memcpy(r.data(), t.data(), PW * sizeof(uint64_t));
    return r;
}

inline const Tables& tables() {
    static Tables T;
    static std::once_flag once;
    std::call_once(once, [] {
        // one output bit of a sequence that has run long enough for any transient to be gone
        HostSequence seq(5489u);
        for (int i = 0; i < 2000; ++i) seq.next();
        std::vector<uint8_t> s(2 * DEG + 128);
        for (auto& b : s) b = (uint8_t)(seq.next() & 1u);
        Poly c{};
        const int L = berlekamp_massey(s, c);
        if (L != DEG) return;
        // characteristic polynomial of the forward shift: phi_j = c_{L-j}
        for (int j = 0; j <= DEG; ++j)
            if (get_bit(c.data(), DEG - j)) flip_bit(T.phi.data(), j);
        if (!get_bit(T.phi.data(), DEG) || !get_bit(T.phi.data(), 0)) return;
        // self-check on fresh bits: XOR_j phi_j s[m + j] = 0
        std::vector<uint8_t> s2(DEG + 64);
        for (auto& b : s2) b = (uint8_t)(seq.next() & 1u);
        for (int m = 0; m < 32; ++m) {
            int acc = 0;
            for (int j = 0; j <= DEG; ++j) acc ^= get_bit(T.phi.data(), j) & s2[m + j];
            if (acc) return;
        }
        T.pow2.resize(MAX_K);
        Poly p{};
        flip_bit(p.data(), 1);   // x
        for (int k = 0; k < MAX_K; ++k) {
            T.pow2[k] = p;
            p = square_mod(p, T.phi);
        }
        T.ok = true;
    });
    return T;
}

// x^(2^k) mod phi as 624 32-bit words (bit i of the polynomial = bit i & 31 of word i >> 5)
inline bool pow2_words(int k, uint32_t* out624) {
    const Tables& T = tables();
    if (!T.ok || k < 0 || k >= MAX_K) return false;
    memcpy(out624, T.pow2[k].data(), 624 * sizeof(uint32_t));
    return true;
}

}  // namespace mtjump
