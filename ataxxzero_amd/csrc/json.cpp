// Finished-game records -> the reference's JSON-lines entry
// (cpp/self_play_client.cpp:512,565-578,639-641): nlohmann::json dumps objects
// with sorted keys and no whitespace, so one game is
//   {"boards":[[49 ints]...],"dists":[{"a7c6":0.0025,...}...],"moves":["b7",...],"result":1}
// boards: index x + 7*y with y = 0 at rank 7, 1 = x, 2 = o, blockers written as 0
// (serialize_board_for_json :88-107); dists: expanded root edges -> visits / total.
#include <algorithm>
#include <charconv>
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

// nlohmann::json 3.x writes a double as the shortest digit string that reads back as the same double (Grisu2), laid out
// by detail::dtoa_impl::format_buffer with min_exp = -4, max_exp = 15: plain decimals while the decimal point lies within
// that many digits of the first digit (0.0001, 0.0025, 1.0), d[.ddd]e-XX with at least two exponent digits outside
// (2.5e-05).  Python's repr(float) lays the values of [0, 1] out the same way (tests/test_json_format.py compares whole
// lines with json.dumps).  std::to_chars alone would pick "5e-04" for 1 / 2000: shorter, and not what the reference writes.
static void append_double(double v, std::string &out)
{
    if (v == 0.0) {
        out += "0.0";
        return;
    }
    if (v < 0.0) {
        out += '-';
        v = -v;
    }
    char tmp[48];
    auto r = std::to_chars(tmp, tmp + sizeof(tmp) - 1, v, std::chars_format::scientific);  // d[.ddd]e[+-]XX, shortest digits
    *r.ptr = 0;
    char digits[24];
    int k = 0;
    const char *p = tmp;
    for (; *p && *p != 'e'; p++)
        if (*p != '.')
            digits[k++] = *p;
    const int n = atoi(p + 1) + 1;  // the decimal point sits after the n-th digit
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

// One record -> its line, for callers that hold records themselves and for the CPU-side test of the format: host code only,
// no device is touched.
extern "C" int azh_format_record_json(const uint32_t *rec, int64_t words, int32_t with_ids, char *buf, int64_t cap,
                                      int64_t *used)
{
    if (!rec || !buf || !used)
        return azh_fail(-1, "azh_format_record_json: null argument");
    *used = 0;
    if (words < 8 || rec[0] != 0x415A4847u /* "AZHG", engine.hip RING_MAGIC */ || rec[5] < 8 || (int64_t)rec[5] > words)
        return azh_fail(-2, "azh_format_record_json: not a finished-game record (%lld words handed over, header says %u)",
                        (long long)words, words >= 8 ? rec[5] : 0u);
    size_t pos = 8;
    for (uint32_t p = 0; p < rec[3]; p++) {
        if (pos + 6 > rec[5] || pos + 6 + (rec[pos + 4] >> 16) > rec[5])
            return azh_fail(-2, "azh_format_record_json: ply %u of %u runs past the record's %u words", p, rec[3], rec[5]);
        pos += 6 + (rec[pos + 4] >> 16);
    }
    const std::string line = azh_format_game_json(rec, rec[5], with_ids != 0);
    *used = (int64_t)line.size();
    if ((int64_t)line.size() > cap)
        return azh_fail(-6, "azh_format_record_json: the line needs %zu bytes, the buffer has %lld", line.size(), (long long)cap);
    memcpy(buf, line.data(), line.size());
    return 0;
}
