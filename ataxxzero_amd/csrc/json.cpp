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

static void append_double(double v, std::string &out)
{
    char tmp[40];
    auto r = std::to_chars(tmp, tmp + sizeof(tmp), v);  // shortest round-trip form
    std::string s(tmp, r.ptr);
    if (s.find_first_of(".en") == std::string::npos)
        s += ".0";  // nlohmann keeps floats recognisable as floats
    out += s;
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
