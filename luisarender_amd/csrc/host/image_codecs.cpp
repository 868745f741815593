// image_codecs.cpp — JPEG, BMP and TGA readers ("image file readers").
//
// The reference loads every LDR texture through stb_image (LoadedImage::load, src/util/imageio.cpp:486-538:
// stbi_info / stbi_load_from_file with 1, 2 or 4 expected channels); src/ext/stb is an empty submodule in the
// snapshot.  What is restated here is the published formats — ITU-T T.81 (baseline, extended-sequential and
// progressive Huffman JPEG, restart intervals, any sampling factors), the BMP and Truevision TGA file layouts —
// with the numerics stb_image uses where the standard leaves a choice, so that texels come out as in the
// reference: the Loeffler-Ligtenberg-Moschytz integer IDCT with 12-bit constants, rounding column pass (>> 10)
// and row pass (>> 17, +128 level shift), "fancy" triangle-filter chroma upsampling (3:1 taps), and the 20-bit
// fixed-point YCbCr -> RGB conversion.  Values are returned in [0, 1] (byte / 255) like the other 8-bit
// readers of image_io.cpp.  Not supported (clear error): arithmetic coding, lossless and 12-bit JPEG,
// CMYK / YCCK, RLE-compressed or 1/4/16-bit BMP.
#include "scene.h"

#include <algorithm>
#include <array>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <string>

namespace lr {

namespace {

namespace fs = std::filesystem;

std::vector<uint8_t> read_file(const fs::path &path) {
    std::ifstream f{path, std::ios::binary};
    if (!f) { throw Error{"Failed to load image '" + path.string() + "'."}; }
    return {std::istreambuf_iterator<char>{f}, std::istreambuf_iterator<char>{}};
}

LoadedImage make_image(uint32_t w, uint32_t h, uint32_t channels) {
    LoadedImage img;
    img.width = w, img.height = h, img.channels = channels;
    img.pixels.assign(static_cast<size_t>(w) * h * 4u, 1.f);
    return img;
}

// ------------------------------------------------------------------ JPEG
constexpr uint8_t zigzag_to_natural[64 + 15] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};// (corrupt run lengths land on the last coefficient)

struct HuffmanTable {
    bool defined{false};
    // canonical code: for each length l, the first code value and the index of its first symbol
    int32_t max_code[18]{};
    int32_t first_index[17]{};
    uint16_t first_code[17]{};
    uint8_t symbols[256]{};
    uint8_t fast_symbol[512]{}, fast_length[512]{};// 9-bit look-ahead
    bool build(const uint8_t counts[16], const uint8_t *syms, uint32_t n) {// false: the lengths do not form a prefix code
        std::memcpy(symbols, syms, n);
        std::memset(fast_length, 0, sizeof(fast_length));
        uint32_t code = 0u, k = 0u;
        for (auto l = 1; l <= 16; l++) {
            first_index[l] = static_cast<int32_t>(k) - static_cast<int32_t>(code);
            first_code[l] = static_cast<uint16_t>(code);
            for (auto i = 0u; i < counts[l - 1]; i++, k++, code++) {
                if (code >= (1u << l)) { return false; }// over-subscribed (corrupt table)
                if (l <= 9) {
                    auto lo = code << (9 - l);
                    for (auto f = 0u; f < (1u << (9 - l)); f++) { fast_symbol[lo + f] = symbols[k], fast_length[lo + f] = static_cast<uint8_t>(l); }
                }
            }
            max_code[l] = static_cast<int32_t>(code) - 1;// -1 when there is no code of this length yet
            if (counts[l - 1] == 0u) { max_code[l] = -1; }
            code <<= 1u;
        }
        max_code[17] = 0x7fffffff;
        defined = true;
        return true;
    }
};

struct JpegComponent {
    uint32_t id{}, h{}, v{}, tq{}, td{}, ta{};
    uint32_t width{}, height{};        // samples of this component inside the image
    uint32_t blocks_w{}, blocks_h{};   // padded to whole MCUs
    int32_t dc_pred{};
    std::vector<int16_t> coefficients; // progressive: [blocks_h][blocks_w][64], natural order
    std::vector<uint8_t> samples;      // [blocks_h * 8][blocks_w * 8]
};

class JpegDecoder {
    const std::vector<uint8_t> &_d;
    const std::string _name;
    size_t _pos{2u};
    uint16_t _quant[4][64]{};// natural order
    HuffmanTable _dc[4], _ac[4];
    std::vector<JpegComponent> _comp;
    uint32_t _width{}, _height{}, _h_max{1u}, _v_max{1u}, _mcus_x{}, _mcus_y{};
    uint32_t _restart_interval{};
    bool _progressive{false}, _have_frame{false};
    int _adobe_transform{-1};
    bool _jfif{false};
    // entropy-coded segment reader
    uint32_t _bits{}, _bit_count{};
    bool _hit_marker{false};
    uint32_t _eob_run{};

    [[noreturn]] void fail(const std::string &why) const { throw Error{"JPEG image '" + _name + "': " + why + "."}; }
    uint8_t u8() { if (_pos >= _d.size()) { fail("truncated file"); } return _d[_pos++]; }
    uint32_t u16() { auto a = u8(); return (static_cast<uint32_t>(a) << 8u) | u8(); }

    void fill_bits() {
        while (_bit_count <= 24u) {
            uint32_t byte = 0u;
            if (!_hit_marker && _pos < _d.size()) {
                byte = _d[_pos];
                if (byte == 0xffu) {
                    auto next = _pos + 1u < _d.size() ? _d[_pos + 1u] : 0xd9u;
                    if (next == 0u) { _pos += 2u; }          // stuffed zero
                    else { _hit_marker = true, byte = 0u; }  // a marker ends the segment: feed zeros
                } else {
                    _pos++;
                }
            }
            _bits |= byte << (24u - _bit_count);
            _bit_count += 8u;
        }
    }
    uint32_t get_bits(uint32_t n) {
        if (n == 0u) { return 0u; }
        if (_bit_count < n) { fill_bits(); }
        auto v = _bits >> (32u - n);
        _bits <<= n, _bit_count -= n;
        return v;
    }
    uint32_t get_bit() { return get_bits(1u); }
    int32_t receive_extend(uint32_t s) {// T.81 F.2.2.1: s magnitude bits, sign by the leading bit
        if (s == 0u) { return 0; }
        auto v = static_cast<int32_t>(get_bits(s));
        return v < (1 << (s - 1u)) ? v - (1 << s) + 1 : v;
    }
    uint32_t decode(const HuffmanTable &t) {
        if (!t.defined) { fail("scan uses an undefined Huffman table"); }
        if (_bit_count < 16u) { fill_bits(); }
        auto look = _bits >> 23u;
        if (auto l = t.fast_length[look]) {
            _bits <<= l, _bit_count -= l;
            return t.fast_symbol[look];
        }
        auto code = static_cast<int32_t>(_bits >> 16u);
        for (auto l = 10; l <= 16; l++) {
            auto c = code >> (16 - l);
            if (t.max_code[l] >= 0 && c <= t.max_code[l] && c >= static_cast<int32_t>(t.first_code[l])) {
                _bits <<= l, _bit_count -= static_cast<uint32_t>(l);
                return t.symbols[(t.first_index[l] + c) & 255];
            }
        }
        fail("corrupt Huffman code");
    }
    void reset_entropy() {
        _bits = 0u, _bit_count = 0u, _hit_marker = false, _eob_run = 0u;
        for (auto &c : _comp) { c.dc_pred = 0; }
    }
    void expect_restart() {// between restart intervals: byte-align, consume RSTn
        _bits = 0u, _bit_count = 0u;
        _hit_marker = false;
        while (_pos + 1u < _d.size() && !(_d[_pos] == 0xffu && _d[_pos + 1u] >= 0xd0u && _d[_pos + 1u] <= 0xd7u)) {
            if (_d[_pos] == 0xffu && _d[_pos + 1u] != 0u && _d[_pos + 1u] != 0xffu) { return; }// some other marker: leave it to the caller
            _pos++;
        }
        if (_pos + 1u < _d.size()) { _pos += 2u; }
        _eob_run = 0u;
        for (auto &c : _comp) { c.dc_pred = 0; }
    }

    // ---- inverse DCT: Loeffler-Ligtenberg-Moschytz, 12-bit fixed-point constants
    // constants are float literals scaled by 2^12, "+ 0.5" and truncated TOWARDS ZERO (so negative ones round up in magnitude - 1)
    static constexpr int32_t fx(float x) { return static_cast<int32_t>(static_cast<double>(x * 4096.f) + 0.5); }
    struct Idct1D { int32_t x0, x1, x2, x3, t0, t1, t2, t3; };
    static Idct1D idct_1d(int32_t s0, int32_t s1, int32_t s2, int32_t s3, int32_t s4, int32_t s5, int32_t s6, int32_t s7) {
        Idct1D r;
        // even part
        auto p1 = (s2 + s6) * fx(0.5411961f);
        auto e2 = p1 + s6 * fx(-1.847759065f);
        auto e3 = p1 + s2 * fx(0.765366865f);
        auto e0 = (s0 + s4) * 4096;
        auto e1 = (s0 - s4) * 4096;
        r.x0 = e0 + e3, r.x3 = e0 - e3, r.x1 = e1 + e2, r.x2 = e1 - e2;
        // odd part
        auto o0 = s7, o1 = s5, o2 = s3, o3 = s1;
        auto q3 = o0 + o2, q4 = o1 + o3, q1 = o0 + o3, q2 = o1 + o2;
        auto q5 = (q3 + q4) * fx(1.175875602f);
        o0 *= fx(0.298631336f), o1 *= fx(2.053119869f), o2 *= fx(3.072711026f), o3 *= fx(1.501321110f);
        q1 = q5 + q1 * fx(-0.899976223f);
        q2 = q5 + q2 * fx(-2.562915447f);
        q3 = q3 * fx(-1.961570560f);
        q4 = q4 * fx(-0.390180644f);
        r.t3 = o3 + q1 + q4, r.t2 = o2 + q2 + q3, r.t1 = o1 + q2 + q4, r.t0 = o0 + q1 + q3;
        return r;
    }
    static uint8_t clamp8(int32_t v) { return static_cast<uint8_t>(v < 0 ? 0 : (v > 255 ? 255 : v)); }
    static void idct_block(const int16_t *in, uint8_t *out, size_t stride) {
        int32_t tmp[64];
        for (auto c = 0; c < 8; c++) {
            auto d = in + c;
            auto v = tmp + c;
            if (d[8] == 0 && d[16] == 0 && d[24] == 0 && d[32] == 0 && d[40] == 0 && d[48] == 0 && d[56] == 0) {
                auto dc = static_cast<int32_t>(d[0]) * 4;
                for (auto r = 0; r < 8; r++) { v[r * 8] = dc; }
                continue;
            }
            auto k = idct_1d(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56]);
            k.x0 += 512, k.x1 += 512, k.x2 += 512, k.x3 += 512;
            v[0] = (k.x0 + k.t3) >> 10, v[56] = (k.x0 - k.t3) >> 10;
            v[8] = (k.x1 + k.t2) >> 10, v[48] = (k.x1 - k.t2) >> 10;
            v[16] = (k.x2 + k.t1) >> 10, v[40] = (k.x2 - k.t1) >> 10;
            v[24] = (k.x3 + k.t0) >> 10, v[32] = (k.x3 - k.t0) >> 10;
        }
        for (auto r = 0; r < 8; r++) {
            auto v = tmp + r * 8;
            auto o = out + static_cast<size_t>(r) * stride;
            auto k = idct_1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            constexpr int32_t bias = 65536 + (128 << 17);// rounding + level shift
            k.x0 += bias, k.x1 += bias, k.x2 += bias, k.x3 += bias;
            o[0] = clamp8((k.x0 + k.t3) >> 17), o[7] = clamp8((k.x0 - k.t3) >> 17);
            o[1] = clamp8((k.x1 + k.t2) >> 17), o[6] = clamp8((k.x1 - k.t2) >> 17);
            o[2] = clamp8((k.x2 + k.t1) >> 17), o[5] = clamp8((k.x2 - k.t1) >> 17);
            o[3] = clamp8((k.x3 + k.t0) >> 17), o[4] = clamp8((k.x3 - k.t0) >> 17);
        }
    }

    // ---- block decoders
    void decode_block_sequential(JpegComponent &c, int16_t data[64]) {
        std::memset(data, 0, 64 * sizeof(int16_t));
        auto t = decode(_dc[c.td]);
        if (t > 15u) { fail("bad DC size"); }
        c.dc_pred += receive_extend(t);
        data[0] = static_cast<int16_t>(c.dc_pred * _quant[c.tq][0]);
        for (auto k = 1u; k < 64u;) {
            auto rs = decode(_ac[c.ta]);
            auto r = rs >> 4u, s = rs & 15u;
            if (s == 0u) {
                if (r != 15u) { break; }// end of block
                k += 16u;
            } else {
                k += r;
                auto zig = zigzag_to_natural[k++];
                data[zig] = static_cast<int16_t>(receive_extend(s) * _quant[c.tq][zig]);
            }
        }
    }
    void decode_block_dc_progressive(JpegComponent &c, int16_t *data, uint32_t ah, uint32_t al) {
        if (ah == 0u) {// first pass
            auto t = decode(_dc[c.td]);
            if (t > 15u) { fail("bad DC size"); }
            c.dc_pred += receive_extend(t);
            data[0] = static_cast<int16_t>(c.dc_pred * (1 << al));
        } else if (get_bit()) {// refinement
            data[0] = static_cast<int16_t>(data[0] + (1 << al));
        }
    }
    void decode_block_ac_progressive(JpegComponent &c, int16_t *data, uint32_t ss, uint32_t se, uint32_t ah, uint32_t al) {
        auto &table = _ac[c.ta];
        if (ah == 0u) {// first pass, T.81 G.1.2.2
            if (_eob_run > 0u) { _eob_run--; return; }
            for (auto k = ss; k <= se;) {
                auto rs = decode(table);
                auto r = rs >> 4u, s = rs & 15u;
                if (s == 0u) {
                    if (r < 15u) {
                        _eob_run = (1u << r) - 1u;
                        if (r) { _eob_run += get_bits(r); }
                        break;
                    }
                    k += 16u;
                } else {
                    k += r;
                    auto zig = zigzag_to_natural[k++];
                    data[zig] = static_cast<int16_t>(receive_extend(s) * (1 << al));
                }
            }
            return;
        }
        // refinement, T.81 G.1.2.3
        const auto bit = static_cast<int16_t>(1 << al);
        auto refine = [&](int16_t &p) {
            if (get_bit() && (p & bit) == 0) { p = static_cast<int16_t>(p > 0 ? p + bit : p - bit); }
        };
        if (_eob_run > 0u) {
            _eob_run--;
            for (auto k = ss; k <= se; k++) {
                auto &p = data[zigzag_to_natural[k]];
                if (p != 0) { refine(p); }
            }
            return;
        }
        auto k = ss;
        do {
            auto rs = decode(table);
            auto r = static_cast<int32_t>(rs >> 4u);
            auto s = static_cast<int32_t>(rs & 15u);
            if (s == 0) {
                if (r < 15) {
                    _eob_run = (1u << r) - 1u;
                    if (r) { _eob_run += get_bits(static_cast<uint32_t>(r)); }
                    r = 64;// run to the end of the band, refining on the way
                }
            } else {
                if (s != 1) { fail("bad refinement code"); }
                s = get_bit() ? bit : -bit;
            }
            while (k <= se) {
                auto &p = data[zigzag_to_natural[k++]];
                if (p != 0) {
                    refine(p);
                } else {
                    if (r == 0) { p = static_cast<int16_t>(s); break; }
                    r--;
                }
            }
        } while (k <= se);
    }

    // ---- markers
    void read_dqt(uint32_t length) {
        auto end = _pos + length;
        while (_pos < end) {
            auto pq_tq = u8();
            uint32_t pq = pq_tq >> 4u, tq = pq_tq & 15u;
            if (pq > 1u || tq > 3u) { fail("bad quantisation table"); }
            for (auto i = 0; i < 64; i++) { _quant[tq][zigzag_to_natural[i]] = static_cast<uint16_t>(pq ? u16() : u8()); }
        }
    }
    void read_dht(uint32_t length) {
        auto end = _pos + length;
        while (_pos < end) {
            auto tc_th = u8();
            uint32_t tc = tc_th >> 4u, th = tc_th & 15u;
            if (tc > 1u || th > 3u) { fail("bad Huffman table"); }
            uint8_t counts[16];
            auto n = 0u;
            for (auto &c : counts) { c = u8(), n += c; }
            if (n > 256u || _pos + n > _d.size()) { fail("bad Huffman table"); }
            if (!(tc ? _ac : _dc)[th].build(counts, _d.data() + _pos, n)) { fail("bad Huffman table"); }
            _pos += n;
        }
    }
    void read_sof(uint32_t marker) {
        if (_have_frame) { fail("more than one frame"); }
        _progressive = marker == 0xc2u;
        if (u8() != 8u) { fail("only 8-bit precision is supported"); }
        _height = u16(), _width = u16();
        if (_width == 0u || _height == 0u) { fail("empty image"); }
        if (static_cast<uint64_t>(_width) * _height > (1ull << 28u)) { fail("image too large"); }
        auto n = u8();
        if (n != 1u && n != 3u) { fail(n == 4u ? "CMYK / YCCK images are not supported" : "bad component count"); }
        _comp.resize(n);
        for (auto &c : _comp) {
            c.id = u8();
            auto hv = u8();
            c.h = hv >> 4u, c.v = hv & 15u, c.tq = u8();
            if (c.h == 0u || c.h > 4u || c.v == 0u || c.v > 4u || c.tq > 3u) { fail("bad component"); }
            _h_max = std::max(_h_max, c.h), _v_max = std::max(_v_max, c.v);
        }
        _mcus_x = (_width + 8u * _h_max - 1u) / (8u * _h_max);
        _mcus_y = (_height + 8u * _v_max - 1u) / (8u * _v_max);
        for (auto &c : _comp) {
            c.width = (_width * c.h + _h_max - 1u) / _h_max;
            c.height = (_height * c.v + _v_max - 1u) / _v_max;
            c.blocks_w = _mcus_x * c.h, c.blocks_h = _mcus_y * c.v;
            c.samples.assign(static_cast<size_t>(c.blocks_w) * c.blocks_h * 64u, 0u);
            if (_progressive) { c.coefficients.assign(static_cast<size_t>(c.blocks_w) * c.blocks_h * 64u, 0); }
        }
        _have_frame = true;
    }
    void read_sos() {
        if (!_have_frame) { fail("scan before frame header"); }
        auto n = u8();
        if (n < 1u || n > _comp.size()) { fail("bad scan component count"); }
        std::vector<JpegComponent *> scan;
        for (auto i = 0u; i < n; i++) {
            auto id = u8();
            auto tables = u8();
            auto it = std::find_if(_comp.begin(), _comp.end(), [&](auto &c) { return c.id == id; });
            if (it == _comp.end()) { fail("scan names an unknown component"); }
            it->td = tables >> 4u, it->ta = tables & 15u;
            if (it->td > 3u || it->ta > 3u) { fail("bad table selector"); }
            scan.push_back(&*it);
        }
        auto ss = static_cast<uint32_t>(u8()), se = static_cast<uint32_t>(u8());
        auto a = u8();
        auto ah = static_cast<uint32_t>(a >> 4u), al = static_cast<uint32_t>(a & 15u);
        if (_progressive) {
            if (ss > 63u || se > 63u || ss > se || ah > 13u || al > 13u || (ss == 0u && se != 0u) || (ss != 0u && n != 1u)) { fail("bad progressive scan parameters"); }
        } else {
            ss = 0u, se = 63u, ah = al = 0u;
        }
        reset_entropy();
        auto decode_one = [&](JpegComponent &c, uint32_t bx, uint32_t by) {
            auto block = (static_cast<size_t>(by) * c.blocks_w + bx) * 64u;
            if (_progressive) {
                if (ss == 0u) { decode_block_dc_progressive(c, c.coefficients.data() + block, ah, al); }
                else { decode_block_ac_progressive(c, c.coefficients.data() + block, ss, se, ah, al); }
            } else {
                int16_t data[64];
                decode_block_sequential(c, data);
                idct_block(data, c.samples.data() + (static_cast<size_t>(by) * 8u * c.blocks_w + bx) * 8u, static_cast<size_t>(c.blocks_w) * 8u);
            }
        };
        auto todo = _restart_interval ? _restart_interval : 0x7fffffffu;
        auto after_mcu = [&](bool last) {
            if (--todo == 0u && !last) {
                expect_restart();
                todo = _restart_interval;
            }
        };
        if (n == 1u) {// non-interleaved: the component's own block grid, unpadded
            auto &c = *scan[0];
            auto w = (c.width + 7u) / 8u, h = (c.height + 7u) / 8u;
            for (auto by = 0u; by < h; by++) {
                for (auto bx = 0u; bx < w; bx++) {
                    decode_one(c, bx, by);
                    after_mcu(by + 1u == h && bx + 1u == w);
                }
            }
        } else {
            for (auto my = 0u; my < _mcus_y; my++) {
                for (auto mx = 0u; mx < _mcus_x; mx++) {
                    for (auto c : scan) {
                        for (auto y = 0u; y < c->v; y++) {
                            for (auto x = 0u; x < c->h; x++) { decode_one(*c, mx * c->h + x, my * c->v + y); }
                        }
                    }
                    after_mcu(my + 1u == _mcus_y && mx + 1u == _mcus_x);
                }
            }
        }
        // skip to the next marker (padding bits / bytes of the entropy-coded segment)
        while (_pos + 1u < _d.size() && !(_d[_pos] == 0xffu && _d[_pos + 1u] != 0u && _d[_pos + 1u] != 0xffu && !(_d[_pos + 1u] >= 0xd0u && _d[_pos + 1u] <= 0xd7u))) { _pos++; }
    }
    void finish_progressive() {
        for (auto &c : _comp) {
            auto w = (c.width + 7u) / 8u, h = (c.height + 7u) / 8u;
            for (auto by = 0u; by < h; by++) {
                for (auto bx = 0u; bx < w; bx++) {
                    auto data = c.coefficients.data() + (static_cast<size_t>(by) * c.blocks_w + bx) * 64u;
                    for (auto i = 0; i < 64; i++) { data[i] = static_cast<int16_t>(data[i] * _quant[c.tq][i]); }
                    idct_block(data, c.samples.data() + (static_cast<size_t>(by) * 8u * c.blocks_w + bx) * 8u, static_cast<size_t>(c.blocks_w) * 8u);
                }
            }
        }
    }

    // ---- upsampling: one output row from the two nearest component rows
    static void upsample_row(uint8_t *out, const uint8_t *near, const uint8_t *far, uint32_t w, uint32_t hs, uint32_t vs) {
        if (hs == 1u && vs == 1u) { std::memcpy(out, near, w); return; }
        if (hs == 1u && vs == 2u) {
            for (auto i = 0u; i < w; i++) { out[i] = static_cast<uint8_t>((3 * near[i] + far[i] + 2) >> 2); }
            return;
        }
        if (hs == 2u && vs == 1u) {
            if (w == 1u) { out[0] = out[1] = near[0]; return; }
            out[0] = near[0];
            out[1] = static_cast<uint8_t>((near[0] * 3 + near[1] + 2) >> 2);
            for (auto i = 1u; i + 1u < w; i++) {
                auto n = 3 * near[i] + 2;
                out[i * 2u] = static_cast<uint8_t>((n + near[i - 1u]) >> 2);
                out[i * 2u + 1u] = static_cast<uint8_t>((n + near[i + 1u]) >> 2);
            }
            out[(w - 1u) * 2u] = static_cast<uint8_t>((near[w - 2u] * 3 + near[w - 1u] + 2) >> 2);
            out[(w - 1u) * 2u + 1u] = near[w - 1u];
            return;
        }
        if (hs == 2u && vs == 2u) {
            auto t1 = 3 * near[0] + far[0];
            if (w == 1u) { out[0] = out[1] = static_cast<uint8_t>((t1 + 2) >> 2); return; }
            out[0] = static_cast<uint8_t>((t1 + 2) >> 2);
            for (auto i = 1u; i < w; i++) {
                auto t0 = t1;
                t1 = 3 * near[i] + far[i];
                out[i * 2u - 1u] = static_cast<uint8_t>((3 * t0 + t1 + 8) >> 4);
                out[i * 2u] = static_cast<uint8_t>((3 * t1 + t0 + 8) >> 4);
            }
            out[w * 2u - 1u] = static_cast<uint8_t>((t1 + 2) >> 2);
            return;
        }
        for (auto i = 0u; i < w; i++) {// other ratios: replicate
            for (auto j = 0u; j < hs; j++) { out[i * hs + j] = near[i]; }
        }
    }

public:
    JpegDecoder(const std::vector<uint8_t> &data, std::string name) : _d{data}, _name{std::move(name)} {}

    LoadedImage run() {
        if (_d.size() < 4u || _d[0] != 0xffu || _d[1] != 0xd8u) { fail("not a JPEG file"); }
        auto done = false;
        while (!done) {
            while (_pos < _d.size() && _d[_pos] != 0xffu) { _pos++; }
            while (_pos < _d.size() && _d[_pos] == 0xffu) { _pos++; }
            if (_pos >= _d.size()) { break; }
            auto marker = static_cast<uint32_t>(_d[_pos++]) | 0u;
            if (marker == 0xd9u) { break; }
            if (marker == 0u || (marker >= 0xd0u && marker <= 0xd7u) || marker == 0x01u) { continue; }
            auto length = u16();
            if (length < 2u || _pos + length - 2u > _d.size()) { fail("bad segment length"); }
            auto next = _pos + length - 2u;
            switch (marker) {
                case 0xc0: case 0xc1: case 0xc2: read_sof(marker); break;
                case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                    fail("lossless, hierarchical and arithmetic-coded JPEG are not supported");
                case 0xc4: read_dht(length - 2u); break;
                case 0xdb: read_dqt(length - 2u); break;
                case 0xdd: _restart_interval = u16(); break;
                case 0xe0: _jfif = length >= 7u && std::memcmp(_d.data() + _pos, "JFIF", 5) == 0; break;
                case 0xee:
                    if (length >= 14u && std::memcmp(_d.data() + _pos, "Adobe", 5) == 0) { _adobe_transform = _d[_pos + 11u]; }
                    break;
                case 0xda:
                    read_sos();// the scan header, then the entropy-coded data up to the next marker
                    next = _pos;
                    break;
                default: break;
            }
            _pos = next;
        }
        if (!_have_frame) { fail("no frame header"); }
        if (_progressive) { finish_progressive(); }
        // ---- to RGB
        auto channels = static_cast<uint32_t>(_comp.size());
        auto img = make_image(_width, _height, channels);
        std::vector<std::vector<uint8_t>> rows(_comp.size());
        for (auto &r : rows) { r.resize(static_cast<size_t>(_mcus_x) * _h_max * 8u + 16u); }
        struct Resample { uint32_t hs, vs, ystep, ypos, w_lores; size_t line0, line1; };
        std::vector<Resample> rs(_comp.size());
        for (size_t k = 0; k < _comp.size(); k++) {
            auto &c = _comp[k];
            rs[k] = {_h_max / c.h, _v_max / c.v, (_v_max / c.v) >> 1u, 0u, (_width + _h_max / c.h - 1u) / (_h_max / c.h), 0u, 0u};
        }
        auto rgb_direct = channels == 3u && (_adobe_transform == 0 || (!_jfif && _adobe_transform < 0 && _comp[0].id == 'R' && _comp[1].id == 'G' && _comp[2].id == 'B'));
        for (auto y = 0u; y < _height; y++) {
            for (size_t k = 0; k < _comp.size(); k++) {
                auto &c = _comp[k];
                auto &r = rs[k];
                auto stride = static_cast<size_t>(c.blocks_w) * 8u;
                auto bottom = r.ystep >= (r.vs >> 1u);
                auto near = c.samples.data() + (bottom ? r.line1 : r.line0) * stride;
                auto far = c.samples.data() + (bottom ? r.line0 : r.line1) * stride;
                upsample_row(rows[k].data(), near, far, r.w_lores, r.hs, r.vs);
                if (++r.ystep >= r.vs) {
                    r.ystep = 0u;
                    r.line0 = r.line1;
                    if (++r.ypos < c.height) { r.line1++; }
                }
            }
            auto dst = img.pixels.data() + static_cast<size_t>(y) * _width * 4u;
            for (auto x = 0u; x < _width; x++, dst += 4) {
                if (channels == 1u) {
                    dst[0] = dst[1] = dst[2] = static_cast<float>(rows[0][x]) / 255.f;
                } else if (rgb_direct) {
                    dst[0] = static_cast<float>(rows[0][x]) / 255.f, dst[1] = static_cast<float>(rows[1][x]) / 255.f, dst[2] = static_cast<float>(rows[2][x]) / 255.f;
                } else {// YCbCr -> RGB in 20-bit fixed point
                    auto fixed = [](float v) { return static_cast<int32_t>(v * 4096.0f + 0.5f) << 8; };
                    auto yy = (static_cast<int32_t>(rows[0][x]) << 20) + (1 << 19);
                    auto cb = static_cast<int32_t>(rows[1][x]) - 128, cr = static_cast<int32_t>(rows[2][x]) - 128;
                    auto r = yy + cr * fixed(1.40200f);
                    auto g = yy - cr * fixed(0.71414f) + static_cast<int32_t>(static_cast<uint32_t>(-cb * fixed(0.34414f)) & 0xffff0000u);
                    auto b = yy + cb * fixed(1.77200f);
                    dst[0] = static_cast<float>(clamp8(r >> 20)) / 255.f, dst[1] = static_cast<float>(clamp8(g >> 20)) / 255.f, dst[2] = static_cast<float>(clamp8(b >> 20)) / 255.f;
                }
                dst[3] = 1.f;
            }
        }
        return img;
    }
};

// ------------------------------------------------------------------ BMP (uncompressed 8-bit palette / 24 / 32 bit)
LoadedImage read_bmp_data(const std::vector<uint8_t> &d, const std::string &name) {
    auto fail = [&](const std::string &why) -> LoadedImage { throw Error{"BMP image '" + name + "': " + why + "."}; };
    auto rd16 = [&](size_t p) { return static_cast<uint32_t>(d[p]) | (static_cast<uint32_t>(d[p + 1u]) << 8u); };
    auto rd32 = [&](size_t p) { return rd16(p) | (rd16(p + 2u) << 16u); };
    if (d.size() < 54u || d[0] != 'B' || d[1] != 'M') { return fail("not a BMP file"); }
    auto offset = rd32(10u), header = rd32(14u);
    if (header < 40u) { return fail("OS/2 headers are not supported"); }
    auto w = static_cast<int32_t>(rd32(18u)), h = static_cast<int32_t>(rd32(22u));
    auto bpp = rd16(28u), compression = rd32(30u);
    auto top_down = h < 0;
    h = std::abs(h);
    if (w <= 0 || h == 0) { return fail("empty image"); }
    if (w > 32768 || h > 32768) { return fail("image too large"); }
    if (!((compression == 0u && (bpp == 8u || bpp == 24u || bpp == 32u)) || (compression == 3u && bpp == 32u))) {
        return fail("only uncompressed 8-bit palette, 24-bit and 32-bit images are supported");
    }
    uint32_t mask[4] = {0x00ff0000u, 0x0000ff00u, 0x000000ffu, 0xff000000u};
    if (compression == 3u) {
        if (d.size() < 70u) { return fail("truncated header"); }
        mask[0] = rd32(54u), mask[1] = rd32(58u), mask[2] = rd32(62u);
        mask[3] = header >= 56u ? rd32(66u) : 0u;
    }
    auto channels = bpp == 32u && mask[3] != 0u ? 4u : 3u;
    auto palette = 14u + header;
    auto row_bytes = ((static_cast<size_t>(w) * bpp + 31u) / 32u) * 4u;
    if (offset + row_bytes * static_cast<size_t>(h) > d.size()) { return fail("truncated pixel data"); }
    if (bpp == 8u && static_cast<size_t>(palette) + 256u * 4u > d.size()) { return fail("truncated palette"); }
    auto img = make_image(static_cast<uint32_t>(w), static_cast<uint32_t>(h), channels);
    auto field = [](uint32_t v, uint32_t m) {// a masked field scaled to 8 bits
        if (m == 0u) { return 255u; }
        auto shift = static_cast<uint32_t>(__builtin_ctz(m));
        auto bits = static_cast<uint32_t>(__builtin_popcount(m));
        auto x = (v & m) >> shift;
        return bits >= 8u ? x >> (bits - 8u) : (x * 255u) / ((1u << bits) - 1u);
    };
    auto any_alpha = false;
    for (auto y = 0; y < h; y++) {
        auto src = d.data() + offset + row_bytes * static_cast<size_t>(top_down ? y : h - 1 - y);
        auto dst = img.pixels.data() + static_cast<size_t>(y) * static_cast<size_t>(w) * 4u;
        for (auto x = 0; x < w; x++, dst += 4) {
            uint32_t r, g, b, a = 255u;
            if (bpp == 8u) {
                auto e = d.data() + palette + static_cast<size_t>(src[x]) * 4u;
                b = e[0], g = e[1], r = e[2];
            } else if (bpp == 24u) {
                b = src[x * 3], g = src[x * 3 + 1], r = src[x * 3 + 2];
            } else {
                auto v = static_cast<uint32_t>(src[x * 4]) | (static_cast<uint32_t>(src[x * 4 + 1]) << 8u) | (static_cast<uint32_t>(src[x * 4 + 2]) << 16u) | (static_cast<uint32_t>(src[x * 4 + 3]) << 24u);
                r = field(v, mask[0]), g = field(v, mask[1]), b = field(v, mask[2]);
                if (channels == 4u) { a = field(v, mask[3]), any_alpha = any_alpha || a != 0u; }
            }
            dst[0] = static_cast<float>(r) / 255.f, dst[1] = static_cast<float>(g) / 255.f, dst[2] = static_cast<float>(b) / 255.f, dst[3] = static_cast<float>(a) / 255.f;
        }
    }
    if (channels == 4u && !any_alpha) {// an all-zero alpha channel means "no alpha" in practice (stb_image does the same)
        for (size_t i = 3u; i < img.pixels.size(); i += 4u) { img.pixels[i] = 1.f; }
    }
    return img;
}

// ------------------------------------------------------------------ TGA (true colour / grey / palette, raw or RLE)
LoadedImage read_tga_data(const std::vector<uint8_t> &d, const std::string &name) {
    auto fail = [&](const std::string &why) -> LoadedImage { throw Error{"TGA image '" + name + "': " + why + "."}; };
    if (d.size() < 18u) { return fail("truncated header"); }
    auto id_length = d[0], map_type = d[1], type = d[2];
    auto map_first = static_cast<uint32_t>(d[3]) | (static_cast<uint32_t>(d[4]) << 8u);
    auto map_length = static_cast<uint32_t>(d[5]) | (static_cast<uint32_t>(d[6]) << 8u);
    auto map_bits = d[7];
    auto w = static_cast<uint32_t>(d[12]) | (static_cast<uint32_t>(d[13]) << 8u), h = static_cast<uint32_t>(d[14]) | (static_cast<uint32_t>(d[15]) << 8u);
    auto bpp = d[16], descriptor = d[17];
    auto rle = (type & 8u) != 0u;
    auto kind = type & 7u;// 1 palette, 2 true colour, 3 grey
    if (w == 0u || h == 0u || kind < 1u || kind > 3u) { return fail("unsupported image type"); }
    auto texel_bits = kind == 1u ? map_bits : bpp;
    if ((kind == 1u && (map_type != 1u || bpp != 8u)) || (kind == 3u && bpp != 8u && bpp != 16u) ||
        (kind != 3u && texel_bits != 15u && texel_bits != 16u && texel_bits != 24u && texel_bits != 32u)) { return fail("unsupported pixel format"); }
    size_t pos = 18u + id_length;
    auto map = d.data() + pos;
    auto map_entry = (static_cast<size_t>(map_bits) + 7u) / 8u;
    if (map_type == 1u) { pos += map_entry * map_length; }
    auto bytes = static_cast<size_t>((bpp + 7u) / 8u);
    std::vector<uint8_t> raw(static_cast<size_t>(w) * h * bytes);
    if (!rle) {
        if (pos + raw.size() > d.size()) { return fail("truncated pixel data"); }
        std::memcpy(raw.data(), d.data() + pos, raw.size());
    } else {
        size_t out = 0u;
        while (out < raw.size()) {
            if (pos >= d.size()) { return fail("truncated RLE data"); }
            auto packet = d[pos++];
            auto count = static_cast<size_t>(packet & 127u) + 1u;
            if (packet & 128u) {
                if (pos + bytes > d.size()) { return fail("truncated RLE data"); }
                for (size_t i = 0; i < count && out < raw.size(); i++, out += bytes) { std::memcpy(raw.data() + out, d.data() + pos, bytes); }
                pos += bytes;
            } else {
                auto n = std::min(count * bytes, raw.size() - out);
                if (pos + n > d.size()) { return fail("truncated RLE data"); }
                std::memcpy(raw.data() + out, d.data() + pos, n);
                pos += n, out += n;
            }
        }
    }
    auto channels = kind == 3u ? (bpp == 16u ? 2u : 1u) : (texel_bits == 32u ? 4u : 3u);
    auto img = make_image(w, h, channels);
    auto top_down = (descriptor & 0x20u) != 0u, right_left = (descriptor & 0x10u) != 0u;
    auto colour = [&](const uint8_t *p, uint32_t bits, float *dst) {
        if (bits == 15u || bits == 16u) {
            auto v = static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8u);
            auto c5 = [](uint32_t x) { return static_cast<float>((x * 255u) / 31u) / 255.f; };
            dst[0] = c5((v >> 10u) & 31u), dst[1] = c5((v >> 5u) & 31u), dst[2] = c5(v & 31u);
        } else {
            dst[0] = static_cast<float>(p[2]) / 255.f, dst[1] = static_cast<float>(p[1]) / 255.f, dst[2] = static_cast<float>(p[0]) / 255.f;
            if (bits == 32u) { dst[3] = static_cast<float>(p[3]) / 255.f; }
        }
    };
    for (auto y = 0u; y < h; y++) {
        auto sy = top_down ? y : h - 1u - y;
        for (auto x = 0u; x < w; x++) {
            auto sx = right_left ? w - 1u - x : x;
            auto p = raw.data() + (static_cast<size_t>(sy) * w + sx) * bytes;
            auto dst = img.pixels.data() + (static_cast<size_t>(y) * w + x) * 4u;
            if (kind == 3u) {
                dst[0] = dst[1] = dst[2] = static_cast<float>(p[0]) / 255.f;
                if (bpp == 16u) { dst[3] = static_cast<float>(p[1]) / 255.f; }
            } else if (kind == 1u) {
                auto index = static_cast<uint32_t>(p[0]);
                index = index >= map_first ? index - map_first : 0u;
                if (index >= map_length) { index = 0u; }
                colour(map + index * map_entry, map_bits, dst);
            } else {
                colour(p, bpp, dst);
            }
        }
    }
    return img;
}

}// namespace

LoadedImage read_jpeg(const std::string &path) {
    auto data = read_file(path);
    return JpegDecoder{data, path}.run();
}

LoadedImage read_bmp(const std::string &path) { return read_bmp_data(read_file(path), path); }

LoadedImage read_tga(const std::string &path) { return read_tga_data(read_file(path), path); }

}// namespace lr
