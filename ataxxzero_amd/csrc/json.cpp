// Finished-game records -> the reference's JSON-lines entry
// (cpp/self_play_client.cpp:512,565-578,639-641): nlohmann::json dumps objects
// with sorted keys and no whitespace, so one game is
//   {"boards":[[49 ints]...],"dists":[{"a7c6":0.0025,...}...],"moves":["b7",...],"result":1}
// boards: index x + 7*y with y = 0 at rank 7, 1 = x, 2 = o, blockers written as 0
// (serialize_board_for_json :88-107); dists: expanded root edges -> visits / total.
#include <algorithm>
#include <charconv>
#include <cmath>
#include <string>
#include <vector>

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "azh_host.h"

static void move_string(uint32_t mv, std::string &out)
{
    const int from = mv & 0xFF, to = (mv >> 8) & 0xFF;
    if (from != to) {  // cpp/move.cpp:11-21
        out.push_back((char)('a' + from % 7));
        out.push_back((char)('1' + from / 7));
    }
    out.push_back((char)('a' + to % 7));
    out.push_back((char)('1' + to / 7));
}

// The digits of a double as nlohmann::json (3.1+) finds them: Grisu2 (F. Loitsch, "Printing floating-point numbers quickly
// and accurately with integers", PLDI 2010) with the 64-bit "do-it-yourself" floats, alpha = -60 / gamma = -32 and cached
// powers of ten in steps of 10^8 that the library uses.  Grisu2 is shortest in ~99.9 % of the cases, not in all: 0.0976823894238616
// is written 0.09768238942386159 — and a `dists` value is written the reference's way only if it is written THAT way, so the
// algorithm is restated here instead of asking std::to_chars for the shortest digits (tests/test_json_format.py compares with
// the library itself, oracle/json_entry_dump.cpp).  Only the two cached powers that are exact in 64 bits are carried (10^4,
// 10^12): they serve every double in [7.3e-12, 32767] — a visit ratio is in [1/60000, 1]; outside, shortest digits.
namespace grisu {

struct Fp {
    uint64_t f;
    int e;
};

static inline Fp mul(Fp x, Fp y)  // upper 64 bits of the product, rounded half up
{
    const unsigned __int128 p = (unsigned __int128)x.f * y.f + ((unsigned __int128)1 << 63);
    return {(uint64_t)(p >> 64), x.e + y.e + 64};
}

static inline Fp normalize(Fp x)
{
    while (!(x.f >> 63)) {
        x.f <<= 1;
        x.e--;
    }
    return x;
}

// digits (no leading zeros) and exponent with value = digits * 10^exponent; false if v is outside the supported range
static bool digits(double v, char *buf, int &len, int &dec_exp)
{
    uint64_t bits;
    memcpy(&bits, &v, 8);
    const uint64_t F = bits & ((1ull << 52) - 1);
    const int E = (int)(bits >> 52) & 0x7FF;
    if (E == 0 || E == 0x7FF)
        return false;
    const Fp val = {F | (1ull << 52), E - 1075};
    // the neighbours' midpoints: every decimal in (m-, m+) reads back as v
    const bool lower_closer = F == 0 && E > 1;
    const Fp mp = {2 * val.f + 1, val.e - 1};
    const Fp mm = lower_closer ? Fp{4 * val.f - 1, val.e - 2} : Fp{2 * val.f - 1, val.e - 1};
    const Fp w_plus = normalize(mp);
    const Fp w_minus = {mm.f << (mm.e - w_plus.e), w_plus.e};
    const Fp w = normalize(val);
    // cached power 10^k with alpha <= e_w + e_c + 64 <= gamma
    const int f = -60 - w_plus.e - 1;
    const int k = (f * 78913) / (1 << 18) + (f > 0 ? 1 : 0);
    const int index = (300 + k + 7) / 8;
    Fp c;
    int ck;
    if (index == 38) {
        c = {0x9C40000000000000ull, -50};  // 10^4
        ck = 4;
    } else if (index == 39) {
        c = {0xE8D4A51000000000ull, -24};  // 10^12
        ck = 12;
    } else {
        return false;
    }
    const Fp W = mul(w, c), Wm = mul(w_minus, c), Wp = mul(w_plus, c);
    // one unit of safety on either side: the products are off by at most one
    const Fp M_minus = {Wm.f + 1, Wm.e}, M_plus = {Wp.f - 1, Wp.e};
    dec_exp = -ck;
    uint64_t delta = M_plus.f - M_minus.f, dist = M_plus.f - W.f;
    const int sh = -M_plus.e;
    const uint64_t one = 1ull << sh;
    uint32_t p1 = (uint32_t)(M_plus.f >> sh);
    uint64_t p2 = M_plus.f & (one - 1);
    auto round_weed = [&](uint64_t rest, uint64_t ten_k) {
        // step the last digit down while that brings the number closer to w without leaving the interval
        while (rest < dist && delta - rest >= ten_k && (rest + ten_k < dist || dist - rest > rest + ten_k - dist)) {
            buf[len - 1]--;
            rest += ten_k;
        }
    };
    uint32_t pow10 = 1;
    int n = 1;
    while (n < 10 && p1 >= pow10 * 10ull) {
        pow10 *= 10;
        n++;
    }
    len = 0;
    while (n > 0) {  // the integral part, digit by digit
        const uint32_t d = p1 / pow10;
        p1 %= pow10;
        buf[len++] = (char)('0' + d);
        n--;
        const uint64_t rest = ((uint64_t)p1 << sh) + p2;
        if (rest <= delta) {
            dec_exp += n;
            round_weed(rest, (uint64_t)pow10 << sh);
            return true;
        }
        pow10 /= 10;
    }
    int m = 0;
    for (;;) {  // the fractional part
        p2 *= 10;
        const uint64_t d = p2 >> sh;
        p2 &= one - 1;
        buf[len++] = (char)('0' + d);
        m++;
        delta *= 10;
        dist *= 10;
        if (p2 <= delta)
            break;
    }
    dec_exp -= m;
    round_weed(p2, one);
    return true;
}

}  // namespace grisu

// nlohmann::json writes a double as the digits above laid out by detail::dtoa_impl::format_buffer with min_exp = -4,
// max_exp = 15: plain decimals while the decimal point lies within that many digits of the first digit (0.0001, 0.0025,
// 1.0), d[.ddd]e-XX with at least two exponent digits outside (2.5e-05).  (std::to_chars alone would pick "5e-04" for
// 1 / 2000: shorter, and not what the reference writes.)
static void append_double(double v, std::string &out)
{
    if (!std::isfinite(v)) {
        out += "null";  // nlohmann's dump_float writes null for NaN and the infinities
        return;
    }
    if (v == 0.0) {
        out += "0.0";
        return;
    }
    if (v < 0.0) {
        out += '-';
        v = -v;
    }
    char digits[40];
    int k = 0, n = 0;  // k digits; the decimal point sits after the n-th
    int dec_exp = 0;
    if (grisu::digits(v, digits, k, dec_exp)) {
        n = k + dec_exp;
    } else {
        char tmp[48];
        auto r = std::to_chars(tmp, tmp + sizeof(tmp) - 1, v, std::chars_format::scientific);  // d[.ddd]e[+-]XX, shortest digits
        *r.ptr = 0;
        const char *p = tmp;
        for (; *p && *p != 'e'; p++)
            if (*p != '.')
                digits[k++] = *p;
        n = (*p == 'e' ? atoi(p + 1) : 0) + 1;
    }
    if (k <= n && n <= 15) {
        out.append(digits, (size_t)k);
        out.append((size_t)(n - k), '0');
        out += ".0";
    } else if (0 < n && n <= 15) {
        out.append(digits, (size_t)n);
        out += '.';
        out.append(digits + n, (size_t)(k - n));
    } else if (-4 < n && n <= 0) {
        out += "0.";
        out.append((size_t)(-n), '0');
        out.append(digits, (size_t)k);
    } else {
        out += digits[0];
        if (k > 1) {
            out += '.';
            out.append(digits + 1, (size_t)(k - 1));
        }
        int e = n - 1;
        out += e < 0 ? "e-" : "e+";
        if (e < 0)
            e = -e;
        if (e < 10)
            out += '0';
        out += std::to_string(e);
    }
}

// rec: ring record (engine.hip k_advance): 8-word header {magic, slot, uid, plies,
// result, words, random_ply + 1 (0 = none), 0} then per ply {x lo, x hi, o lo, o hi, move | nd << 16, 0,
// nd x (move | visits << 16)}.
// with_ids (arena): two extra keys, "slot" and "uid", so the caller can tell which net had x.
std::string azh_format_game_json(const uint32_t *rec, size_t words, bool with_ids)
{
    const uint32_t plies = rec[3], result = rec[4];
    std::string boards = "[", dists = "[", moves = "[";
    size_t pos = 8;
    for (uint32_t p = 0; p < plies && pos + 6 <= words; p++) {
        const uint64_t x = (uint64_t)rec[pos] | ((uint64_t)rec[pos + 1] << 32);
        const uint64_t o = (uint64_t)rec[pos + 2] | ((uint64_t)rec[pos + 3] << 32);
        const uint32_t mv = rec[pos + 4] & 0xFFFFu, nd = rec[pos + 4] >> 16;
        if (p) {
            boards += ',';
            dists += ',';
            moves += ',';
        }
        boards += '[';
        for (int y = 0; y < 7; y++)
            for (int xx = 0; xx < 7; xx++) {
                const uint64_t m = 1ULL << (xx + 7 * (6 - y));
                if (y || xx)
                    boards += ',';
                boards += (x & m) ? '1' : ((o & m) ? '2' : '0');
            }
        boards += ']';
        moves += '"';
        move_string(mv, moves);
        moves += '"';
        uint64_t total = 0;
        std::vector<std::pair<std::string, uint32_t>> ents;
        for (uint32_t j = 0; j < nd && pos + 6 + j < words; j++) {
            const uint32_t e = rec[pos + 6 + j];
            std::string key;
            move_string(e & 0xFFFFu, key);
            ents.emplace_back(key, e >> 16);
            total += e >> 16;
        }
        std::sort(ents.begin(), ents.end());
        dists += '{';
        for (size_t j = 0; j < ents.size(); j++) {
            if (j)
                dists += ',';
            dists += '"';
            dists += ents[j].first;
            dists += "\":";
            append_double(total ? (double)ents[j].second / (double)total : 0.0, dists);
        }
        dists += '}';
        pos += 6 + nd;
    }
    std::string out = "{\"boards\":" + boards + "],\"dists\":" + dists + "],\"moves\":" + moves + "],\"result\":";
    if (rec[6])  // ONE_RANDOM_MOVE games (cpp/self_play_client.cpp:517); keys stay sorted as nlohmann emits them
        out = out.substr(0, out.size() - 9) + "\"random_ply\":" + std::to_string(rec[6] - 1) + ",\"result\":";
    out += std::to_string(result);
    if (with_ids)
        out += ",\"slot\":" + std::to_string(rec[1]) + ",\"uid\":" + std::to_string(rec[2]);
    out += '}';
    return out;
}

// Is what starts at `rec` (at most `avail` words of it readable) a whole, well-formed record?  The ply structure must
// walk to the record's end exactly.  Used by azh_format_record_json and by the engine's drain when it looks for the next
// header after a damaged record: a payload word may equal the magic (board halves are arbitrary bit patterns), and only a
// header whose record passes this walk is taken for one.
bool azh_record_well_formed(const uint32_t *rec, size_t avail, uint32_t max_plies, const char **why)
{
    const char *dummy;
    if (!why)
        why = &dummy;
    *why = "not a finished-game record";
    if (avail < 8 || rec[0] != 0x415A4847u /* "AZHG", engine.hip RING_MAGIC */ || rec[5] < 8 || (size_t)rec[5] > avail)
        return false;
    if (rec[7] == 1)  // the marker a dropped game leaves: a header and nothing else
        return rec[5] == 8;
    *why = "header fields out of range";
    if (rec[7] > 2 || rec[4] > 2 || (max_plies && rec[3] > max_plies))
        return false;
    *why = "a ply runs past the record's words";
    size_t pos = 8;
    for (uint32_t p = 0; p < rec[3]; p++) {
        if (pos + 6 > rec[5])
            return false;
        const uint32_t nd = rec[pos + 4] >> 16;
        if (nd > 256 || pos + 6 + nd > rec[5])
            return false;
        pos += 6 + nd;
    }
    *why = "the plies do not fill the record";
    return pos == rec[5];
}

// One record -> its line, for callers that hold records themselves and for the CPU-side test of the format: host code only,
// no device is touched.
extern "C" int azh_format_record_json(const uint32_t *rec, int64_t words, int32_t with_ids, char *buf, int64_t cap,
                                      int64_t *used)
{
    if (!rec || !buf || !used)
        return azh_fail(-1, "azh_format_record_json: null argument");
    *used = 0;
    const char *why = nullptr;
    if (words < 8 || !azh_record_well_formed(rec, (size_t)words, 0u, &why))
        return azh_fail(-2, "azh_format_record_json: %s (%lld words handed over, header says %u words, %u plies)",
                        why ? why : "not a finished-game record", (long long)words, words >= 8 ? rec[5] : 0u,
                        words >= 8 ? rec[3] : 0u);
    if (rec[7] == 1)
        return azh_fail(-2, "azh_format_record_json: the record is the marker of a dropped game, it has no line");
    const std::string line = azh_format_game_json(rec, rec[5], with_ids != 0);
    *used = (int64_t)line.size();
    if ((int64_t)line.size() > cap)
        return azh_fail(-6, "azh_format_record_json: the line needs %zu bytes, the buffer has %lld", line.size(), (long long)cap);
    memcpy(buf, line.data(), line.size());
    return 0;
}
