// st_jpeg.h — JPEG (ITU-T T.81) decoder for glTF textures: baseline, extended-sequential and progressive DCT, 8-bit,
// Huffman-coded, greyscale or three components with any sampling factors, restart intervals. Included by st_gltf.cpp only.
//
// glTF allows image/jpeg next to image/png; Bevy decodes both through the `image` crate before bevy-strolle ever sees the
// pixels (bevy-strolle/src/stages/prepare.rs:182-260 receives raw RGBA). The arithmetic after entropy decoding follows
// the algorithms every mainstream decoder shares, so that textures come out as they do elsewhere: the 13-bit fixed-point
// "slow" inverse DCT (Loeffler-Ligtenberg-Moshovitz), triangle-filter chroma upsampling for 2x1 and 2x2 subsampling, and
// the 16-bit fixed-point YCbCr -> RGB conversion of JFIF. Not handled (reported, not guessed): arithmetic coding, lossless
// and hierarchical modes, 12-bit samples, four-component (CMYK / YCCK) files.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace st_jpeg {

struct Error {
    bool unsupported;
    const char* message;
};
[[noreturn]] inline void fail(const char* m) { throw Error{false, m}; }
[[noreturn]] inline void unsupported(const char* m) { throw Error{true, m}; }

struct Image {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba;
};

class Decoder {
  public:
    Decoder(const uint8_t* data, size_t size, uint32_t max_side) : p_(data), n_(size), max_side_(max_side) {}

    Image decode() {
        if (n_ < 4 || p_[0] != 0xFF || p_[1] != 0xD8) fail("JPEG: no SOI marker");
        pos_ = 2;
        bool done = false;
        while (!done) {
            const int marker = next_marker();
            switch (marker) {
                case 0xC0: case 0xC1: case 0xC2: frame_header(marker); break;
                case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xCB: case 0xCD: case 0xCE: case 0xCF: unsupported("JPEG: lossless / hierarchical mode");
                case 0xC9: case 0xCA: unsupported("JPEG: arithmetic coding");
                case 0xC4: huffman_tables(); break;
                case 0xDB: quantization_tables(); break;
                case 0xDD: { const size_t len = segment_length(); if (len != 2) fail("JPEG: bad DRI"); restart_interval_ = be16(p_ + pos_); pos_ += len; break; }
                case 0xDA: scan(); break;
                case 0xD9: done = true; break;
                case 0xEE: adobe_marker(); break;
                default:
                    if (marker >= 0xD0 && marker <= 0xD7) break;  // stray restart marker
                    pos_ += segment_length();                      // APPn, COM, ...
            }
        }
        if (!have_frame_) fail("JPEG: no frame header");
        if (!any_scan_) fail("JPEG: no scan");
        return reconstruct();
    }

  private:
    struct Huffman {
        bool present = false;
        uint8_t values[256];
        int32_t maxcode[18];   // largest code of each length (-1: none)
        int32_t valptr[17];    // index into values of the first code of each length
        int32_t mincode[17];
    };
    struct Component {
        int id = 0, h = 1, v = 1, tq = 0;
        int td = 0, ta = 0;                 // tables selected by the current scan
        uint32_t blocks_w = 0, blocks_h = 0;    // block grid padded to whole MCUs
        uint32_t width = 0, height = 0;         // samples that carry picture (ceil(image * h / hmax))
        std::vector<int16_t> coef;              // blocks_w * blocks_h * 64, natural (de-zigzagged) order, not yet dequantized
        int pred = 0;
    };

    const uint8_t* p_;
    size_t n_, pos_ = 0;
    uint32_t max_side_;
    uint16_t qt_[4][64];
    bool have_qt_[4] = {false, false, false, false};
    Huffman dc_[4], ac_[4];
    Component comp_[3];
    int ncomp_ = 0, hmax_ = 1, vmax_ = 1;
    uint32_t width_ = 0, height_ = 0, mcus_x_ = 0, mcus_y_ = 0;
    bool have_frame_ = false, progressive_ = false, any_scan_ = false;
    uint32_t restart_interval_ = 0;
    int adobe_transform_ = -1;
    // entropy-coded segment reader
    uint32_t bitbuf_ = 0;
    int bitcnt_ = 0;
    bool hit_marker_ = false;
    uint32_t eobrun_ = 0;

    static uint32_t be16(const uint8_t* q) { return ((uint32_t)q[0] << 8) | q[1]; }
    size_t segment_length() {
        if (n_ - pos_ < 2) fail("JPEG: truncated segment");
        const size_t len = be16(p_ + pos_);
        if (len < 2 || len > n_ - pos_) fail("JPEG: segment runs past the end of the file");
        pos_ += 2;
        return len - 2;
    }
    int next_marker() {
        while (pos_ < n_ && p_[pos_] != 0xFF) pos_++;      // tolerate garbage between segments
        while (pos_ < n_ && p_[pos_] == 0xFF) pos_++;      // fill bytes
        if (pos_ >= n_) fail("JPEG: file ends before EOI");
        return p_[pos_++];
    }

    static const uint8_t* zigzag() {
        static const uint8_t z[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                      35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        return z;
    }

    void quantization_tables() {
        size_t len = segment_length();
        const uint8_t* q = p_ + pos_;
        pos_ += len;
        while (len > 0) {
            const int precision = q[0] >> 4, id = q[0] & 15;
            if (id > 3 || precision > 1) fail("JPEG: bad DQT");
            const size_t need = 1 + (precision ? 128 : 64);
            if (len < need) fail("JPEG: truncated DQT");
            for (int i = 0; i < 64; i++) qt_[id][zigzag()[i]] = precision ? (uint16_t)be16(q + 1 + 2 * i) : q[1 + i];
            have_qt_[id] = true;
            q += need; len -= need;
        }
    }
    void huffman_tables() {
        size_t len = segment_length();
        const uint8_t* q = p_ + pos_;
        pos_ += len;
        while (len > 0) {
            if (len < 17) fail("JPEG: truncated DHT");
            const int cls = q[0] >> 4, id = q[0] & 15;
            if (cls > 1 || id > 3) fail("JPEG: bad DHT");
            int total = 0;
            for (int i = 1; i <= 16; i++) total += q[i];
            if (total > 256 || len < (size_t)17 + total) fail("JPEG: truncated DHT");
            Huffman& h = cls ? ac_[id] : dc_[id];
            memcpy(h.values, q + 17, (size_t)total);
            int code = 0, k = 0;
            for (int l = 1; l <= 16; l++) {
                h.valptr[l] = k; h.mincode[l] = code;
                k += q[l]; code += q[l];
                h.maxcode[l] = q[l] ? code - 1 : -1;
                if (code > (1 << l)) fail("JPEG: over-subscribed Huffman table");
                code <<= 1;
            }
            h.maxcode[17] = 0x7fffffff;
            h.present = true;
            q += 17 + total; len -= 17 + (size_t)total;
        }
    }
    void adobe_marker() {
        const size_t len = segment_length();
        if (len >= 12 && memcmp(p_ + pos_, "Adobe", 5) == 0) adobe_transform_ = p_[pos_ + 11];
        pos_ += len;
    }
    void frame_header(int marker) {
        const size_t len = segment_length();
        const uint8_t* q = p_ + pos_;
        pos_ += len;
        if (have_frame_) fail("JPEG: second frame header");
        if (len < 6) fail("JPEG: truncated SOF");
        if (q[0] != 8) unsupported("JPEG: sample precision other than 8 bits");
        height_ = be16(q + 1); width_ = be16(q + 3); ncomp_ = q[5];
        if (ncomp_ == 4) unsupported("JPEG: four-component (CMYK / YCCK) image");
        if (ncomp_ != 1 && ncomp_ != 3) fail("JPEG: component count");
        if (width_ == 0 || height_ == 0) unsupported("JPEG: image height given by a DNL marker");
        if (width_ > max_side_ || height_ > max_side_) unsupported("JPEG: image larger than the atlas");
        if (len < (size_t)6 + 3 * ncomp_) fail("JPEG: truncated SOF");
        progressive_ = marker == 0xC2;
        for (int c = 0; c < ncomp_; c++) {
            Component& k = comp_[c];
            k.id = q[6 + 3 * c]; k.h = q[7 + 3 * c] >> 4; k.v = q[7 + 3 * c] & 15; k.tq = q[8 + 3 * c];
            if (k.h < 1 || k.h > 4 || k.v < 1 || k.v > 4 || k.tq > 3) fail("JPEG: bad component parameters");
            if (k.h > hmax_) hmax_ = k.h;
            if (k.v > vmax_) vmax_ = k.v;
        }
        if (ncomp_ == 1) { comp_[0].h = comp_[0].v = 1; hmax_ = vmax_ = 1; }  // a single component is never interleaved
        mcus_x_ = (width_ + 8 * hmax_ - 1) / (8 * hmax_);
        mcus_y_ = (height_ + 8 * vmax_ - 1) / (8 * vmax_);
        for (int c = 0; c < ncomp_; c++) {
            Component& k = comp_[c];
            k.blocks_w = mcus_x_ * k.h; k.blocks_h = mcus_y_ * k.v;
            k.width = (width_ * k.h + hmax_ - 1) / hmax_; k.height = (height_ * k.v + vmax_ - 1) / vmax_;
            k.coef.assign((size_t)k.blocks_w * k.blocks_h * 64, 0);
        }
        have_frame_ = true;
    }

    // ---- entropy-coded data
    void fill_bits() {
        while (bitcnt_ <= 24) {
            uint32_t byte = 0;
            if (!hit_marker_ && pos_ < n_) {
                byte = p_[pos_];
                if (byte == 0xFF) {
                    const uint8_t next = pos_ + 1 < n_ ? p_[pos_ + 1] : 0xD9;
                    if (next == 0) pos_ += 2;                // stuffed zero
                    else { hit_marker_ = true; byte = 0; }  // a marker ends the segment: feed zeros from here on
                } else pos_++;
            } else hit_marker_ = true;
            bitbuf_ |= byte << (24 - bitcnt_);
            bitcnt_ += 8;
        }
    }
    int get_bits(int n) {
        if (n == 0) return 0;
        if (bitcnt_ < n) fill_bits();
        const int v = (int)(bitbuf_ >> (32 - n));
        bitbuf_ <<= n; bitcnt_ -= n;
        return v;
    }
    int get_bit() { return get_bits(1); }
    int decode_symbol(const Huffman& h) {
        if (!h.present) fail("JPEG: scan refers to a Huffman table that was never defined");
        int code = 0;
        for (int l = 1; l <= 16; l++) {
            code = (code << 1) | get_bit();
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.values[h.valptr[l] + code - h.mincode[l]];
        }
        fail("JPEG: invalid Huffman code");
    }
    static int bounded(int pred) { return pred < -(1 << 20) ? -(1 << 20) : pred > (1 << 20) ? (1 << 20) : pred; }  // valid streams stay within 12 bits
    static int extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }
    void restart() {
        // byte-align, expect RSTn
        bitbuf_ = 0; bitcnt_ = 0;
        if (hit_marker_) {
            if (pos_ + 1 < n_ && p_[pos_] == 0xFF && p_[pos_ + 1] >= 0xD0 && p_[pos_ + 1] <= 0xD7) { pos_ += 2; hit_marker_ = false; }
        } else {
            // the encoder's padding bits were not all consumed: look for the marker just ahead
            size_t q = pos_;
            while (q + 1 < n_ && !(p_[q] == 0xFF && p_[q + 1] >= 0xD0 && p_[q + 1] <= 0xD7) && q < pos_ + 4) q++;
            if (q + 1 < n_ && p_[q] == 0xFF && p_[q + 1] >= 0xD0 && p_[q + 1] <= 0xD7) pos_ = q + 2;
        }
        for (int c = 0; c < ncomp_; c++) comp_[c].pred = 0;
        eobrun_ = 0;
    }

    void block_baseline(Component& k, int16_t* b) {
        const int s = decode_symbol(dc_[k.td]);
        if (s > 11) fail("JPEG: bad DC size");
        k.pred = bounded(k.pred + extend(get_bits(s), s));
        b[0] = (int16_t)k.pred;
        for (int i = 1; i < 64;) {
            const int rs = decode_symbol(ac_[k.ta]), r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (r != 15) break;  // EOB
                i += 16;
                continue;
            }
            i += r;
            if (i > 63) fail("JPEG: coefficient index past 63");
            b[zigzag()[i]] = (int16_t)extend(get_bits(sz), sz);
            i++;
        }
    }
    void block_dc_first(Component& k, int16_t* b, int al) {
        const int s = decode_symbol(dc_[k.td]);
        if (s > 11) fail("JPEG: bad DC size");
        k.pred = bounded(k.pred + extend(get_bits(s), s));
        b[0] = (int16_t)(k.pred * (1 << al));
    }
    void block_dc_refine(int16_t* b, int al) {
        if (get_bit()) b[0] = (int16_t)(b[0] | (1 << al));
    }
    void block_ac_first(Component& k, int16_t* b, int ss, int se, int al) {
        if (eobrun_ > 0) { eobrun_--; return; }
        for (int i = ss; i <= se;) {
            const int rs = decode_symbol(ac_[k.ta]), r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (r < 15) {
                    eobrun_ = (1u << r) - 1u;
                    if (r) eobrun_ += (uint32_t)get_bits(r);
                    break;
                }
                i += 16;
                continue;
            }
            i += r;
            if (i > se) fail("JPEG: coefficient index past the band");
            b[zigzag()[i]] = (int16_t)(extend(get_bits(sz), sz) * (1 << al));
            i++;
        }
    }
    void block_ac_refine(Component& k, int16_t* b, int ss, int se, int al) {
        const int p1 = 1 << al, m1 = -(1 << al);
        int i = ss;
        if (eobrun_ == 0) {
            for (; i <= se; i++) {
                const int rs = decode_symbol(ac_[k.ta]);
                int r = rs >> 4;
                const int sz = rs & 15;
                int value = 0;
                if (sz) {
                    if (sz != 1) fail("JPEG: bad refinement size");
                    value = get_bit() ? p1 : m1;
                } else if (r != 15) {
                    eobrun_ = 1u << r;
                    if (r) eobrun_ += (uint32_t)get_bits(r);
                    break;
                }
                // skip r zero-history coefficients, refining the non-zero ones passed on the way
                for (; i <= se; i++) {
                    int16_t& c = b[zigzag()[i]];
                    if (c != 0) {
                        if (get_bit() && (c & p1) == 0) c = (int16_t)(c >= 0 ? c + p1 : c + m1);
                    } else {
                        if (--r < 0) break;
                    }
                }
                if (value) {
                    if (i > se) fail("JPEG: refinement coefficient past the band");
                    b[zigzag()[i]] = (int16_t)value;
                }
            }
        }
        if (eobrun_ > 0) {
            for (; i <= se; i++) {
                int16_t& c = b[zigzag()[i]];
                if (c != 0 && get_bit() && (c & p1) == 0) c = (int16_t)(c >= 0 ? c + p1 : c + m1);
            }
            eobrun_--;
        }
    }

    void scan() {
        if (!have_frame_) fail("JPEG: scan before the frame header");
        const size_t len = segment_length();
        const uint8_t* q = p_ + pos_;
        pos_ += len;
        if (len < 1) fail("JPEG: truncated SOS");
        const int ns = q[0];
        if (ns < 1 || ns > ncomp_ || len < (size_t)4 + 2 * ns) fail("JPEG: bad SOS");
        Component* in_scan[3];
        for (int i = 0; i < ns; i++) {
            Component* k = nullptr;
            for (int c = 0; c < ncomp_; c++) if (comp_[c].id == q[1 + 2 * i]) k = &comp_[c];
            if (!k) fail("JPEG: scan names an unknown component");
            for (int j = 0; j < i; j++) if (in_scan[j] == k) fail("JPEG: component twice in one scan");
            k->td = q[2 + 2 * i] >> 4; k->ta = q[2 + 2 * i] & 15;
            if (k->td > 3 || k->ta > 3) fail("JPEG: bad table selector");
            in_scan[i] = k;
        }
        const int ss = q[1 + 2 * ns], se = q[2 + 2 * ns], ah = q[3 + 2 * ns] >> 4, al = q[3 + 2 * ns] & 15;
        if (progressive_) {
            if (ss > se || se > 63 || al > 13 || ah > 13 || (ss == 0 && se != 0) || (ss > 0 && ns != 1)) fail("JPEG: bad progressive scan parameters");
        } else if (ss != 0 || se != 63 || ah != 0 || al != 0) fail("JPEG: bad sequential scan parameters");
        any_scan_ = true;
        bitbuf_ = 0; bitcnt_ = 0; hit_marker_ = false; eobrun_ = 0;
        for (int c = 0; c < ncomp_; c++) comp_[c].pred = 0;

        auto one_block = [&](Component& k, uint32_t bx, uint32_t by) {
            int16_t* b = &k.coef[((size_t)by * k.blocks_w + bx) * 64];
            if (!progressive_) block_baseline(k, b);
            else if (ss == 0) { if (ah == 0) block_dc_first(k, b, al); else block_dc_refine(b, al); }
            else { if (ah == 0) block_ac_first(k, b, ss, se, al); else block_ac_refine(k, b, ss, se, al); }
        };
        uint32_t until_restart = restart_interval_;
        auto count_mcu = [&](bool last) {
            if (restart_interval_ && !last && --until_restart == 0) { restart(); until_restart = restart_interval_; }
        };
        if (ns == 1) {  // non-interleaved: the component's own block grid, without MCU padding
            Component& k = *in_scan[0];
            const uint32_t bw = (k.width + 7) / 8, bh = (k.height + 7) / 8;
            for (uint32_t by = 0; by < bh; by++)
                for (uint32_t bx = 0; bx < bw; bx++) { one_block(k, bx, by); count_mcu(by + 1 == bh && bx + 1 == bw); }
        } else {
            for (uint32_t my = 0; my < mcus_y_; my++)
                for (uint32_t mx = 0; mx < mcus_x_; mx++) {
                    for (int i = 0; i < ns; i++) {
                        Component& k = *in_scan[i];
                        for (int v = 0; v < k.v; v++)
                            for (int h = 0; h < k.h; h++) one_block(k, mx * k.h + h, my * k.v + v);
                    }
                    count_mcu(my + 1 == mcus_y_ && mx + 1 == mcus_x_);
                }
        }
        // leave the reader at the marker that ended the segment
        if (!hit_marker_) {
            while (pos_ + 1 < n_ && !(p_[pos_] == 0xFF && p_[pos_ + 1] != 0 && !(p_[pos_ + 1] >= 0xD0 && p_[pos_ + 1] <= 0xD7))) pos_++;
        }
    }

    // ---- reconstruction
    static int descale(int64_t x, int n) { return (int)((x + ((int64_t)1 << (n - 1))) >> n); }
    static uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

    // 13-bit fixed-point inverse DCT (the "slow integer" one): columns into a workspace, then rows
    static void idct(const int16_t* in, const uint16_t* q, uint8_t* out, size_t out_stride) {
        constexpr int CB = 13, P1 = 2;
        constexpr int64_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069,
                          F2053 = 16819, F2562 = 20995, F3072 = 25172;
        int ws[64];
        for (int c = 0; c < 8; c++) {
            const int16_t* i = in + c;
            const uint16_t* qq = q + c;
            int* w = ws + c;
            if (!i[8] && !i[16] && !i[24] && !i[32] && !i[40] && !i[48] && !i[56]) {
                const int dc = (int)i[0] * qq[0] * (1 << P1);
                for (int r = 0; r < 8; r++) w[8 * r] = dc;
                continue;
            }
            int64_t z2 = (int64_t)i[16] * qq[16], z3 = (int64_t)i[48] * qq[48];
            int64_t z1 = (z2 + z3) * F0541;
            int64_t tmp2 = z1 + z3 * -F1847, tmp3 = z1 + z2 * F0765;
            z2 = (int64_t)i[0] * qq[0]; z3 = (int64_t)i[32] * qq[32];
            int64_t tmp0 = (z2 + z3) * (1 << CB), tmp1 = (z2 - z3) * (1 << CB);
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = (int64_t)i[56] * qq[56]; tmp1 = (int64_t)i[40] * qq[40]; tmp2 = (int64_t)i[24] * qq[24]; tmp3 = (int64_t)i[8] * qq[8];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
            int64_t z4 = tmp1 + tmp3;
            const int64_t z5 = (z3 + z4) * F1175;
            tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
            z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            w[0] = descale(tmp10 + tmp3, CB - P1); w[56] = descale(tmp10 - tmp3, CB - P1);
            w[8] = descale(tmp11 + tmp2, CB - P1); w[48] = descale(tmp11 - tmp2, CB - P1);
            w[16] = descale(tmp12 + tmp1, CB - P1); w[40] = descale(tmp12 - tmp1, CB - P1);
            w[24] = descale(tmp13 + tmp0, CB - P1); w[32] = descale(tmp13 - tmp0, CB - P1);
        }
        for (int r = 0; r < 8; r++) {
            const int* w = ws + 8 * r;
            uint8_t* o = out + (size_t)r * out_stride;
            int64_t z2 = w[2], z3 = w[6];
            int64_t z1 = (z2 + z3) * F0541;
            int64_t tmp2 = z1 + z3 * -F1847, tmp3 = z1 + z2 * F0765;
            int64_t tmp0 = ((int64_t)w[0] + w[4]) * (1 << CB), tmp1 = ((int64_t)w[0] - w[4]) * (1 << CB);
            const int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
            int64_t z4 = tmp1 + tmp3;
            const int64_t z5 = (z3 + z4) * F1175;
            tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
            z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            constexpr int S = CB + P1 + 3;
            o[0] = clamp8(descale(tmp10 + tmp3, S) + 128); o[7] = clamp8(descale(tmp10 - tmp3, S) + 128);
            o[1] = clamp8(descale(tmp11 + tmp2, S) + 128); o[6] = clamp8(descale(tmp11 - tmp2, S) + 128);
            o[2] = clamp8(descale(tmp12 + tmp1, S) + 128); o[5] = clamp8(descale(tmp12 - tmp1, S) + 128);
            o[3] = clamp8(descale(tmp13 + tmp0, S) + 128); o[4] = clamp8(descale(tmp13 - tmp0, S) + 128);
        }
    }

    struct Plane {
        uint32_t w = 0, h = 0;       // allocated (padded) size
        std::vector<uint8_t> px;
    };

    // full-resolution plane of component c: triangle filter for 2:1 steps (what "fancy upsampling" means), replication otherwise
    Plane upsample(const Component& k, const Plane& src) const {
        const int fx = hmax_ / k.h, fy = vmax_ / k.v;
        if (hmax_ % k.h || vmax_ % k.v) unsupported("JPEG: fractional sampling ratio");
        Plane out;
        out.w = mcus_x_ * 8 * hmax_; out.h = mcus_y_ * 8 * vmax_;
        out.px.assign((size_t)out.w * out.h, 0);
        const uint32_t cw = k.width, ch = k.height;  // samples that carry picture; the filter replicates their edge
        if (fx == 1 && fy == 1) { out.px = src.px; return out; }
        const bool triangle = cw > 2;  // narrower components are replicated (the filter needs a neighbour on both sides somewhere)
        if (fx == 2 && fy == 1 && triangle) {
            for (uint32_t y = 0; y < ch; y++) {
                const uint8_t* in = &src.px[(size_t)y * src.w];
                uint8_t* o = &out.px[(size_t)y * out.w];
                o[0] = in[0]; o[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
                for (uint32_t x = 1; x + 1 < cw; x++) {
                    o[2 * x] = (uint8_t)((in[x] * 3 + in[x - 1] + 1) >> 2);
                    o[2 * x + 1] = (uint8_t)((in[x] * 3 + in[x + 1] + 2) >> 2);
                }
                o[2 * cw - 2] = (uint8_t)((in[cw - 1] * 3 + in[cw - 2] + 1) >> 2); o[2 * cw - 1] = in[cw - 1];
            }
            return out;
        }
        if (fx == 2 && fy == 2 && triangle) {
            for (uint32_t y = 0; y < ch; y++) {
                const uint8_t* near = &src.px[(size_t)y * src.w];
                for (int half = 0; half < 2; half++) {
                    const uint32_t fy_row = half == 0 ? (y == 0 ? 0 : y - 1) : (y + 1 < ch ? y + 1 : ch - 1);
                    const uint8_t* far = &src.px[(size_t)fy_row * src.w];
                    uint8_t* o = &out.px[((size_t)2 * y + half) * out.w];
                    auto colsum = [&](uint32_t x) { return (int)near[x] * 3 + (int)far[x]; };
                    int last = colsum(0), cur = colsum(0), next = colsum(1);
                    o[0] = (uint8_t)((cur * 4 + 8) >> 4); o[1] = (uint8_t)((cur * 3 + next + 7) >> 4);
                    for (uint32_t x = 1; x + 1 < cw; x++) {
                        last = cur; cur = next; next = colsum(x + 1);
                        o[2 * x] = (uint8_t)((cur * 3 + last + 8) >> 4);
                        o[2 * x + 1] = (uint8_t)((cur * 3 + next + 7) >> 4);
                    }
                    last = cur; cur = next;
                    o[2 * cw - 2] = (uint8_t)((cur * 3 + last + 8) >> 4); o[2 * cw - 1] = (uint8_t)((cur * 4 + 7) >> 4);
                }
            }
            return out;
        }
        for (uint32_t y = 0; y < ch * (uint32_t)fy && y < out.h; y++)
            for (uint32_t x = 0; x < cw * (uint32_t)fx && x < out.w; x++) out.px[(size_t)y * out.w + x] = src.px[(size_t)(y / fy) * src.w + x / fx];
        return out;
    }

    Image reconstruct() {
        Plane full[3];
        for (int c = 0; c < ncomp_; c++) {
            Component& k = comp_[c];
            if (!have_qt_[k.tq]) fail("JPEG: component refers to a quantization table that was never defined");
            Plane raw;
            raw.w = k.blocks_w * 8; raw.h = k.blocks_h * 8;
            raw.px.assign((size_t)raw.w * raw.h, 0);
            for (uint32_t by = 0; by < k.blocks_h; by++)
                for (uint32_t bx = 0; bx < k.blocks_w; bx++)
                    idct(&k.coef[((size_t)by * k.blocks_w + bx) * 64], qt_[k.tq], &raw.px[((size_t)by * 8) * raw.w + (size_t)bx * 8], raw.w);
            full[c] = upsample(k, raw);
        }
        Image img;
        img.width = width_; img.height = height_;
        img.rgba.resize((size_t)width_ * height_ * 4);
        const bool rgb = ncomp_ == 3 && (adobe_transform_ == 0 || (adobe_transform_ < 0 && comp_[0].id == 'R' && comp_[1].id == 'G' && comp_[2].id == 'B'));
        constexpr int HALF = 1 << 15;
        for (uint32_t y = 0; y < height_; y++)
            for (uint32_t x = 0; x < width_; x++) {
                uint8_t* o = &img.rgba[((size_t)y * width_ + x) * 4];
                const int Y = full[0].px[(size_t)y * full[0].w + x];
                if (ncomp_ == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; o[3] = 255; continue; }
                const int B = full[1].px[(size_t)y * full[1].w + x], R = full[2].px[(size_t)y * full[2].w + x];
                if (rgb) { o[0] = (uint8_t)Y; o[1] = (uint8_t)B; o[2] = (uint8_t)R; o[3] = 255; continue; }
                const int cb = B - 128, cr = R - 128;
                // 16-bit fixed point: 1.40200, 1.77200, 0.71414, 0.34414 (JFIF)
                const int r = Y + ((91881 * cr + HALF) >> 16);
                const int b = Y + ((116130 * cb + HALF) >> 16);
                const int g = Y + ((-46802 * cr - 22554 * cb + HALF) >> 16);
                o[0] = clamp8(r); o[1] = clamp8(g); o[2] = clamp8(b); o[3] = 255;
            }
        return img;
    }
};

}  // namespace st_jpeg
