// JPEG (baseline and progressive DCT, Huffman coded, 8 bits per sample) to 8-bit RGB.
//
// The reference reads its textures and environment maps with stb_image 2.x
// (framework/src/image.cpp:22, framework/third_party/stb/include/stb/stb_image.h), whose JPEG
// path fixes every choice the standard leaves to a decoder.  Texels must come out with the
// reference's bytes, so those choices are restated here:
//   * the inverse DCT is the integer "slow" form of the IJG code with 12-bit constants, two
//     extra bits kept between the column and the row pass (stb_image.h:2392-2490);
//   * chroma is upsampled with the triangle filter (3:1 weights, centred on the JFIF sample
//     positions) for factors of two, by repetition for any other factor (stb_image.h:3400-3601),
//     the vertical neighbour chosen by stb_image's row counter (stb_image.h:3845-3887);
//   * YCbCr -> RGB in 20-bit fixed point with the green chroma term cut to 16 bits
//     (stb_image.h:3605-3629); components named 'R','G','B', or an Adobe segment with
//     transform 0 and no JFIF segment, are taken as RGB; four components as CMYK / YCCK.
// Huffman decoding, the progressive scans (spectral selection, successive approximation) and
// restart intervals follow ITU-T T.81 (Annex F, G), as stb_image.h:2182-2379,2915-3037 does.
#include "scene_io.h"

#include <array>
#include <cstdint>
#include <cstring>

namespace rt {
namespace {

constexpr int kZigzag[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63
};

// ---- canonical Huffman table (T.81 Annex C), with a 9-bit first-level lookup ----------
struct Huffman {
    bool defined = false;
    int minCode[18] = {}, maxCode[18] = {}, firstIndex[18] = {}; // per code length 1..16
    uint8_t symbols[256] = {};
    uint16_t fast[512] = {}; // (length << 8) | symbol for codes of at most 9 bits, 0 otherwise

    void build(const int counts[16], const uint8_t* values, int n, const std::string& file)
    {
        // the table is checked as a whole before anything is written: a length with more codes
        // than its bits can tell apart would index past the lookup table below
        for (int len = 1, code = 0; len <= 16; len++, code <<= 1) {
            code += counts[len - 1];
            if (code > (1 << len))
                throw RtIoError(file + ": JPEG Huffman table has too many codes of one length");
        }
        std::memcpy(symbols, values, size_t(n));
        std::memset(fast, 0, sizeof(fast));
        int code = 0, index = 0;
        for (int len = 1; len <= 16; len++) {
            minCode[len] = code;
            firstIndex[len] = index;
            for (int k = 0; k < counts[len - 1]; k++, code++, index++) {
                if (len <= 9) {
                    const int base = code << (9 - len);
                    for (int fill = 0; fill < (1 << (9 - len)); fill++)
                        fast[base + fill] = uint16_t((len << 8) | symbols[index]);
                }
            }
            maxCode[len] = code - 1; // -1 below minCode when the length is unused
            code <<= 1;
        }
        defined = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, dcTable = 0, acTable = 0;
    int dcPred = 0;
    int width = 0, height = 0;   // samples that carry picture
    int stride = 0, rows = 0;    // samples stored (whole MCUs)
    std::vector<uint8_t> plane;  // decoded samples
    std::vector<int16_t> coeff;  // progressive: coefficients of every stored block, natural order
};

class JpegDecoder {
public:
    JpegDecoder(const std::vector<unsigned char>& bytes, const std::string& file)
        : m_p(bytes.data()), m_end(bytes.data() + bytes.size()), m_file(file)
    {
    }

    Image decode()
    {
        if (nextMarker() != 0xD8)
            fail("no start-of-image marker");
        bool haveFrame = false;
        for (;;) {
            const int m = nextMarker();
            if (m == 0xD9)
                break;
            if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                if (haveFrame)
                    fail("more than one frame header");
                m_progressive = m == 0xC2;
                frameHeader();
                haveFrame = true;
            } else if (m == 0xDA) {
                if (!haveFrame)
                    fail("scan before the frame header");
                scanHeader();
                scan();
            } else if (m == 0xDC) { // number of lines, must repeat the frame header's
                const int len = get16();
                const int lines = get16();
                if (len != 4 || lines != m_height)
                    fail("bad DNL segment");
            } else {
                segment(m);
            }
        }
        if (!haveFrame)
            fail("no frame header");
        if (m_progressive)
            finishProgressive();
        return toRgb();
    }

private:
    [[noreturn]] void fail(const std::string& what) const { throw RtIoError(m_file + ": JPEG: " + what); }

    // ---- bytes and markers ---------------------------------------------------------
    int get8()
    {
        if (m_p >= m_end)
            fail("file ends inside the data");
        return *m_p++;
    }
    int get16()
    {
        const int hi = get8();
        return (hi << 8) | get8();
    }
    // the next marker: one left over by the entropy-coded data, else 0xFF fill bytes and the code
    int nextMarker()
    {
        if (m_marker >= 0) {
            const int m = m_marker;
            m_marker = -1;
            return m;
        }
        // bytes that are no marker are skipped (padding after a segment or after the scan's data)
        for (;;) {
            int b = get8();
            if (b != 0xFF)
                continue;
            while (b == 0xFF)
                b = get8();
            if (b != 0)
                return b;
        }
    }

    // ---- tables and headers --------------------------------------------------------
    void segment(int m)
    {
        if (m == 0xDD) {
            if (get16() != 4)
                fail("bad DRI segment");
            m_restartInterval = get16();
            return;
        }
        if (m == 0xDB) {
            int left = get16() - 2;
            while (left > 0) {
                const int q = get8();
                const int wide = q >> 4, t = q & 15;
                if (wide > 1 || t > 3)
                    fail("bad quantization table header");
                for (int i = 0; i < 64; i++)
                    m_quant[t][kZigzag[i]] = uint16_t(wide ? get16() : get8());
                left -= wide ? 129 : 65;
            }
            if (left != 0)
                fail("bad DQT segment length");
            return;
        }
        if (m == 0xC4) {
            int left = get16() - 2;
            while (left > 0) {
                const int q = get8();
                const int cls = q >> 4, t = q & 15;
                if (cls > 1 || t > 3)
                    fail("bad Huffman table header");
                int counts[16], n = 0;
                for (int& c : counts)
                    n += c = get8();
                if (n > 256)
                    fail("Huffman table with more than 256 codes");
                uint8_t values[256];
                for (int i = 0; i < n; i++)
                    values[i] = uint8_t(get8());
                (cls ? m_ac[t] : m_dc[t]).build(counts, values, n, m_file);
                left -= 17 + n;
            }
            if (left != 0)
                fail("bad DHT segment length");
            return;
        }
        if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
            int left = get16() - 2;
            if (left < 0)
                fail("bad application segment length");
            if (m == 0xE0 && left >= 5) {
                static const char tag[5] = { 'J', 'F', 'I', 'F', 0 };
                bool ok = true;
                for (char c : tag)
                    ok &= get8() == uint8_t(c);
                left -= 5;
                if (ok)
                    m_jfif = true;
            } else if (m == 0xEE && left >= 12) {
                static const char tag[6] = { 'A', 'd', 'o', 'b', 'e', 0 };
                bool ok = true;
                for (char c : tag)
                    ok &= get8() == uint8_t(c);
                left -= 6;
                if (ok) {
                    get8(), get16(), get16(); // version, flags
                    m_adobeTransform = get8();
                    left -= 6;
                }
            }
            if (m_end - m_p < left)
                fail("file ends inside a segment");
            m_p += left;
            return;
        }
        fail("unsupported marker (only Huffman-coded baseline and progressive DCT frames are read)");
    }

    void frameHeader()
    {
        const int len = get16();
        if (get8() != 8)
            fail("only 8 bits per sample are supported");
        m_height = get16();
        m_width = get16();
        if (m_height == 0 || m_width == 0)
            fail("empty picture");
        const int n = get8();
        if (n != 1 && n != 3 && n != 4)
            fail("bad component count");
        if (len != 8 + 3 * n)
            fail("bad frame header length");
        m_comps.assign(size_t(n), Component {});
        m_rgbIds = 0;
        static const char rgb[3] = { 'R', 'G', 'B' };
        for (int i = 0; i < n; i++) {
            Component& c = m_comps[size_t(i)];
            c.id = get8();
            if (n == 3 && c.id == rgb[i])
                m_rgbIds++;
            const int q = get8();
            c.h = q >> 4, c.v = q & 15;
            c.tq = get8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3)
                fail("bad sampling factors or table number");
            m_hMax = std::max(m_hMax, c.h);
            m_vMax = std::max(m_vMax, c.v);
        }
        for (const Component& c : m_comps)
            if (m_hMax % c.h || m_vMax % c.v)
                fail("sampling factors that are no whole ratios are not supported");
        m_mcusX = (m_width + 8 * m_hMax - 1) / (8 * m_hMax);
        m_mcusY = (m_height + 8 * m_vMax - 1) / (8 * m_vMax);
        for (Component& c : m_comps) {
            c.width = (m_width * c.h + m_hMax - 1) / m_hMax;
            c.height = (m_height * c.v + m_vMax - 1) / m_vMax;
            c.stride = m_mcusX * c.h * 8;
            c.rows = m_mcusY * c.v * 8;
            c.plane.assign(size_t(c.stride) * size_t(c.rows), 0);
            if (m_progressive)
                c.coeff.assign(size_t(c.stride) * size_t(c.rows), 0);
        }
    }

    void scanHeader()
    {
        const int len = get16();
        const int n = get8();
        if (n < 1 || n > 4 || n > int(m_comps.size()) || len != 6 + 2 * n)
            fail("bad scan header");
        m_scan.clear();
        for (int i = 0; i < n; i++) {
            const int id = get8(), q = get8();
            size_t which = 0;
            while (which < m_comps.size() && m_comps[which].id != id)
                which++;
            if (which == m_comps.size() || (q >> 4) > 3 || (q & 15) > 3)
                fail("scan names an unknown component or table");
            m_comps[which].dcTable = q >> 4;
            m_comps[which].acTable = q & 15;
            m_scan.push_back(int(which));
        }
        m_specStart = get8();
        m_specEnd = get8();
        const int a = get8();
        m_succHigh = a >> 4, m_succLow = a & 15;
        if (m_progressive) {
            if (m_specStart > 63 || m_specEnd > 63 || m_specStart > m_specEnd || m_succHigh > 13 || m_succLow > 13)
                fail("bad progressive scan parameters");
            if (m_specStart == 0 ? m_specEnd != 0 : n != 1)
                fail("a progressive scan holds either DC terms or the AC terms of one component");
        } else {
            if (m_specStart != 0 || a != 0)
                fail("bad scan parameters");
            m_specEnd = 63;
        }
    }

    // ---- entropy-coded bits --------------------------------------------------------
    // Up to 25 more bits in m_acc (left-aligned); once a marker turns up the data has ended
    // and zeros follow.
    void fill()
    {
        while (m_bits <= 24) {
            int b = 0;
            if (!m_dataEnded) {
                b = m_p < m_end ? *m_p++ : 0;
                if (b == 0xFF) {
                    int c = m_p < m_end ? *m_p++ : 0xD9;
                    while (c == 0xFF)
                        c = m_p < m_end ? *m_p++ : 0xD9;
                    if (c != 0) {
                        m_marker = c;
                        m_dataEnded = true;
                        b = 0;
                    }
                }
            }
            m_acc |= uint32_t(b) << (24 - m_bits);
            m_bits += 8;
        }
    }
    int bits(int n)
    {
        if (n == 0)
            return 0;
        if (m_bits < n)
            fill();
        const int v = int(m_acc >> (32 - n));
        m_acc <<= n;
        m_bits -= n;
        return v;
    }
    int bit() { return bits(1); }
    // T.81 F.2.2.1: n received bits as a signed value
    int extend(int n)
    {
        if (n == 0)
            return 0;
        const int v = bits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    int symbol(const Huffman& h)
    {
        if (!h.defined)
            fail("scan uses a Huffman table that was never defined");
        if (m_bits < 16)
            fill();
        const uint16_t f = h.fast[m_acc >> 23];
        if (f) {
            const int len = f >> 8;
            m_acc <<= len;
            m_bits -= len;
            return f & 255;
        }
        for (int len = 10; len <= 16; len++) {
            const int code = int(m_acc >> (32 - len));
            if (code <= h.maxCode[len] && code >= h.minCode[len]) {
                m_acc <<= len;
                m_bits -= len;
                return h.symbols[h.firstIndex[len] + code - h.minCode[len]];
            }
        }
        fail("bad Huffman code");
    }
    void resetEntropy()
    {
        m_acc = 0;
        m_bits = 0;
        m_dataEnded = false;
        m_marker = -1;
        m_eobRun = 0;
        for (Component& c : m_comps)
            c.dcPred = 0;
        m_untilRestart = m_restartInterval ? m_restartInterval : 0x7fffffff;
    }
    // after each MCU; false when the data stops where a restart marker was due
    bool restartIfDue()
    {
        if (--m_untilRestart > 0)
            return true;
        if (m_bits < 24)
            fill();
        if (m_marker < 0xD0 || m_marker > 0xD7)
            return false;
        resetEntropy();
        return true;
    }

    // ---- blocks --------------------------------------------------------------------
    // sequential: one block's coefficients, dequantized, natural order
    void sequentialBlock(Component& c, int16_t out[64])
    {
        const uint16_t* q = m_quant[c.tq];
        std::memset(out, 0, 64 * sizeof(int16_t));
        const int t = symbol(m_dc[c.dcTable]);
        if (t > 15)
            fail("bad DC code");
        c.dcPred += extend(t);
        out[0] = int16_t(c.dcPred * q[0]);
        const Huffman& ac = m_ac[c.acTable];
        for (int k = 1; k < 64;) {
            const int rs = symbol(ac);
            const int run = rs >> 4, size = rs & 15;
            if (size == 0) {
                if (rs != 0xF0)
                    break;
                k += 16;
                continue;
            }
            k += run;
            if (k > 63)
                fail("coefficient run leaves the block");
            const int where = kZigzag[k++];
            out[where] = int16_t(extend(size) * q[where]);
        }
    }

    void progressiveDc(Component& c, int16_t* blk)
    {
        if (m_succHigh == 0) {
            const int t = symbol(m_dc[c.dcTable]);
            if (t > 15)
                fail("bad DC code");
            c.dcPred += extend(t);
            blk[0] = int16_t(c.dcPred * (1 << m_succLow));
        } else if (bit()) {
            blk[0] = int16_t(blk[0] + (1 << m_succLow));
        }
    }

    // one more bit of a coefficient that is already non-zero (T.81 G.1.2.3)
    void refine(int16_t& v, int16_t step)
    {
        if (bit() && (v & step) == 0)
            v = int16_t(v > 0 ? v + step : v - step);
    }

    void progressiveAc(Component& c, int16_t* blk)
    {
        const Huffman& ac = m_ac[c.acTable];
        if (m_succHigh == 0) {
            if (m_eobRun) {
                m_eobRun--;
                return;
            }
            for (int k = m_specStart; k <= m_specEnd;) {
                const int rs = symbol(ac);
                const int run = rs >> 4, size = rs & 15;
                if (size == 0) {
                    if (run < 15) { // end of band for 2^run + (run more bits) blocks, this one included
                        m_eobRun = (1 << run) + bits(run) - 1;
                        break;
                    }
                    k += 16;
                    continue;
                }
                k += run;
                if (k > 63)
                    fail("coefficient run leaves the block");
                blk[kZigzag[k++]] = int16_t(extend(size) * (1 << m_succLow));
            }
            return;
        }
        const int16_t step = int16_t(1 << m_succLow);
        int k = m_specStart;
        if (m_eobRun == 0) {
            while (k <= m_specEnd) {
                const int rs = symbol(ac);
                int run = rs >> 4;
                const int size = rs & 15;
                int16_t fresh = 0;
                if (size == 0) {
                    if (run < 15) {
                        m_eobRun = (1 << run) + bits(run); // this block's remainder is handled below
                        break;
                    }
                    // run == 15: sixteen zeros, i.e. fifteen skipped and a zero "placed" on the sixteenth
                } else {
                    if (size != 1)
                        fail("bad code in a refinement scan");
                    fresh = bit() ? step : int16_t(-step);
                }
                while (k <= m_specEnd) {
                    int16_t& v = blk[kZigzag[k++]];
                    if (v != 0) {
                        refine(v, step);
                    } else if (run == 0) {
                        v = fresh;
                        break;
                    } else {
                        run--;
                    }
                }
            }
        }
        if (m_eobRun) { // the rest of the band only refines what is already there
            for (; k <= m_specEnd; k++) {
                int16_t& v = blk[kZigzag[k]];
                if (v != 0)
                    refine(v, step);
            }
            m_eobRun--;
        }
    }

    // jidctint's "slow" integer inverse DCT, constants to 12 bits; in[] natural order, out 8 rows
    static void idct(const int16_t in[64], uint8_t* out, int stride)
    {
        struct Pass {
            int64_t e[4], o[4]; // sample k = e[k] + o[k], sample 7-k = e[k] - o[k]
        };
        auto pass = [](int64_t s0, int64_t s1, int64_t s2, int64_t s3, int64_t s4, int64_t s5, int64_t s6, int64_t s7) {
            constexpr auto fix = [](float x) { return int64_t(int(x * 4096 + 0.5)); };
            Pass r;
            const int64_t z = (s2 + s6) * fix(0.5411961f);
            const int64_t even2 = z + s6 * fix(-1.847759065f), even3 = z + s2 * fix(0.765366865f);
            const int64_t sum = (s0 + s4) * 4096, diff = (s0 - s4) * 4096;
            r.e[0] = sum + even3, r.e[3] = sum - even3, r.e[1] = diff + even2, r.e[2] = diff - even2;
            const int64_t a = s7 + s3, b = s5 + s1, c = s7 + s1, d = s5 + s3;
            const int64_t all = (a + b) * fix(1.175875602f);
            const int64_t pc = all + c * fix(-0.899976223f), pd = all + d * fix(-2.562915447f);
            const int64_t pa = a * fix(-1.961570560f), pb = b * fix(-0.390180644f);
            r.o[0] = s1 * fix(1.501321110f) + pc + pb;
            r.o[1] = s3 * fix(3.072711026f) + pd + pa;
            r.o[2] = s5 * fix(2.053119869f) + pd + pb;
            r.o[3] = s7 * fix(0.298631336f) + pc + pa;
            return r;
        };
        int64_t mid[64];
        for (int x = 0; x < 8; x++) {
            const int16_t* d = in + x;
            const Pass r = pass(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56]);
            for (int k = 0; k < 4; k++) {
                mid[8 * k + x] = (r.e[k] + 512 + r.o[k]) >> 10;
                mid[8 * (7 - k) + x] = (r.e[k] + 512 - r.o[k]) >> 10;
            }
        }
        for (int y = 0; y < 8; y++) {
            const int64_t* v = mid + 8 * y;
            const Pass r = pass(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            uint8_t* o = out + size_t(y) * size_t(stride);
            constexpr int64_t bias = 65536 + (128 << 17); // rounding, and the level shift back to 0..255
            for (int k = 0; k < 4; k++) {
                o[k] = clamp8((r.e[k] + bias + r.o[k]) >> 17);
                o[7 - k] = clamp8((r.e[k] + bias - r.o[k]) >> 17);
            }
        }
    }
    static uint8_t clamp8(int64_t v) { return uint8_t(v < 0 ? 0 : v > 255 ? 255 : v); }

    // ---- one scan ------------------------------------------------------------------
    void scan()
    {
        resetEntropy();
        int16_t block[64];
        bool more = true;
        if (m_scan.size() == 1) {
            // one component: its blocks in raster order, only those that carry picture
            Component& c = m_comps[size_t(m_scan[0])];
            const int bw = (c.width + 7) >> 3, bh = (c.height + 7) >> 3;
            for (int by = 0; by < bh && more; by++)
                for (int bx = 0; bx < bw && more; bx++) {
                    decodeBlock(c, bx, by, block);
                    more = restartIfDue();
                }
        } else {
            for (int my = 0; my < m_mcusY && more; my++)
                for (int mx = 0; mx < m_mcusX && more; mx++) {
                    for (int which : m_scan) {
                        Component& c = m_comps[size_t(which)];
                        for (int y = 0; y < c.v; y++)
                            for (int x = 0; x < c.h; x++)
                                decodeBlock(c, mx * c.h + x, my * c.v + y, block);
                    }
                    more = restartIfDue();
                }
        }
        // bytes between the data and the next marker are skipped by nextMarker()
    }

    void decodeBlock(Component& c, int bx, int by, int16_t scratch[64])
    {
        if (!m_progressive) {
            sequentialBlock(c, scratch);
            idct(scratch, &c.plane[size_t(by) * 8 * size_t(c.stride) + size_t(bx) * 8], c.stride);
            return;
        }
        int16_t* blk = &c.coeff[64 * (size_t(by) * size_t(c.stride / 8) + size_t(bx))];
        if (m_specStart == 0)
            progressiveDc(c, blk);
        else
            progressiveAc(c, blk);
    }

    void finishProgressive()
    {
        int16_t block[64];
        for (Component& c : m_comps) {
            const uint16_t* q = m_quant[c.tq];
            const int bw = (c.width + 7) >> 3, bh = (c.height + 7) >> 3;
            for (int by = 0; by < bh; by++)
                for (int bx = 0; bx < bw; bx++) {
                    const int16_t* blk = &c.coeff[64 * (size_t(by) * size_t(c.stride / 8) + size_t(bx))];
                    for (int i = 0; i < 64; i++)
                        block[i] = int16_t(blk[i] * q[i]);
                    idct(block, &c.plane[size_t(by) * 8 * size_t(c.stride) + size_t(bx) * 8], c.stride);
                }
        }
    }

    // ---- upsampling and colour ---------------------------------------------------------
    static void upsampleRow(uint8_t* out, const uint8_t* near, const uint8_t* far, int w, int hs, int vs)
    {
        if (hs == 1 && vs == 2) {
            for (int i = 0; i < w; i++)
                out[i] = uint8_t((3 * near[i] + far[i] + 2) >> 2);
        } else if (hs == 2 && vs == 1) {
            if (w == 1) {
                out[0] = out[1] = near[0];
                return;
            }
            out[0] = near[0];
            out[1] = uint8_t((near[0] * 3 + near[1] + 2) >> 2);
            for (int i = 1; i < w - 1; i++) {
                const int n = 3 * near[i] + 2;
                out[2 * i] = uint8_t((n + near[i - 1]) >> 2);
                out[2 * i + 1] = uint8_t((n + near[i + 1]) >> 2);
            }
            out[2 * w - 2] = uint8_t((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
            out[2 * w - 1] = near[w - 1];
        } else if (hs == 2 && vs == 2) {
            int cur = 3 * near[0] + far[0];
            out[0] = uint8_t((cur + 2) >> 2);
            if (w == 1) {
                out[1] = out[0];
                return;
            }
            for (int i = 1; i < w; i++) {
                const int prev = cur;
                cur = 3 * near[i] + far[i];
                out[2 * i - 1] = uint8_t((3 * prev + cur + 8) >> 4);
                out[2 * i] = uint8_t((3 * cur + prev + 8) >> 4);
            }
            out[2 * w - 1] = uint8_t((cur + 2) >> 2);
        } else {
            for (int i = 0; i < w; i++)
                for (int j = 0; j < hs; j++)
                    out[i * hs + j] = near[i];
        }
    }

    static uint8_t scale8(int x, int y) // x * y / 255, rounded
    {
        const unsigned t = unsigned(x * y + 128);
        return uint8_t((t + (t >> 8)) >> 8);
    }

    Image toRgb()
    {
        const int n = int(m_comps.size());
        const bool isRgb = n == 3 && (m_rgbIds == 3 || (m_adobeTransform == 0 && !m_jfif));
        struct Cursor {
            int hs, vs, step, row, wLow;
            const uint8_t *line0, *line1;
            std::vector<uint8_t> buffer;
            const uint8_t* out;
        };
        std::vector<Cursor> cur(static_cast<size_t>(n));
        for (int k = 0; k < n; k++) {
            const Component& c = m_comps[size_t(k)];
            Cursor& r = cur[size_t(k)];
            r.hs = m_hMax / c.h, r.vs = m_vMax / c.v;
            r.step = r.vs >> 1;
            r.row = 0;
            r.wLow = (m_width + r.hs - 1) / r.hs;
            r.line0 = r.line1 = c.plane.data();
            r.buffer.assign(size_t(r.wLow) * size_t(r.hs) + 8, 0);
        }
        constexpr auto fixed = [](float x) { return int(x * 4096.0f + 0.5f) << 8; };
        Image img;
        img.width = m_width;
        img.height = m_height;
        img.pixels.reserve(size_t(m_width) * size_t(m_height));
        for (int y = 0; y < m_height; y++) {
            for (int k = 0; k < n; k++) {
                const Component& c = m_comps[size_t(k)];
                Cursor& r = cur[size_t(k)];
                const bool lower = r.step >= (r.vs >> 1); // which stored row is the nearer one
                const uint8_t* near = lower ? r.line1 : r.line0;
                const uint8_t* far = lower ? r.line0 : r.line1;
                if (r.hs == 1 && r.vs == 1) {
                    r.out = near;
                } else {
                    upsampleRow(r.buffer.data(), near, far, r.wLow, r.hs, r.vs);
                    r.out = r.buffer.data();
                }
                if (++r.step >= r.vs) {
                    r.step = 0;
                    r.line0 = r.line1;
                    if (++r.row < c.height)
                        r.line1 += c.stride;
                }
            }
            for (int x = 0; x < m_width; x++) {
                int R, G, B;
                const int c0 = cur[0].out[x];
                if (n == 1) {
                    R = G = B = c0;
                } else if (isRgb || (n == 4 && m_adobeTransform == 0)) {
                    R = c0, G = cur[1].out[x], B = cur[2].out[x];
                } else {
                    const int lum = (c0 << 20) + (1 << 19);
                    const int cb = cur[1].out[x] - 128, cr = cur[2].out[x] - 128;
                    R = (lum + cr * fixed(1.40200f)) >> 20;
                    G = int(lum + cr * -fixed(0.71414f) + int(unsigned(cb * -fixed(0.34414f)) & 0xffff0000u)) >> 20;
                    B = (lum + cb * fixed(1.77200f)) >> 20;
                    R = R < 0 ? 0 : R > 255 ? 255 : R;
                    G = G < 0 ? 0 : G > 255 ? 255 : G;
                    B = B < 0 ? 0 : B > 255 ? 255 : B;
                }
                if (n == 4) { // CMYK (transform 0), YCCK (transform 2); any other value: the fourth channel is ignored
                    const int k4 = cur[3].out[x];
                    if (m_adobeTransform == 0)
                        R = scale8(R, k4), G = scale8(G, k4), B = scale8(B, k4);
                    else if (m_adobeTransform == 2)
                        R = scale8(255 - R, k4), G = scale8(255 - G, k4), B = scale8(255 - B, k4);
                }
                img.pixels.emplace_back(float(R) / 255.0f, float(G) / 255.0f, float(B) / 255.0f);
            }
        }
        return img;
    }

    const unsigned char* m_p;
    const unsigned char* m_end;
    std::string m_file;
    int m_marker = -1;
    bool m_progressive = false, m_jfif = false;
    int m_adobeTransform = -1, m_rgbIds = 0;
    int m_width = 0, m_height = 0, m_hMax = 1, m_vMax = 1, m_mcusX = 0, m_mcusY = 0;
    int m_restartInterval = 0, m_untilRestart = 0;
    std::vector<Component> m_comps;
    std::vector<int> m_scan;
    int m_specStart = 0, m_specEnd = 63, m_succHigh = 0, m_succLow = 0, m_eobRun = 0;
    uint16_t m_quant[4][64] = {};
    Huffman m_dc[4], m_ac[4];
    uint32_t m_acc = 0;
    int m_bits = 0;
    bool m_dataEnded = false;
};

} // namespace

Image decodeJpeg(const std::vector<unsigned char>& bytes, const std::string& file)
{
    return JpegDecoder(bytes, file).decode();
}

} // namespace rt
