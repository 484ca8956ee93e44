// JPEG decoder (baseline and progressive) for map_Kd textures (the reference reads them through stb_image's stbi_loadf, model.cpp:8-23; stb is a
// third-party dependency that is not part of this build).  Written from ITU-T T.81: sequential DCT (SOF0 / SOF1) and progressive DCT
// (SOF2: spectral selection + successive approximation, Annex G -- coefficients are collected over all scans, then transformed), 8-bit
// samples, Huffman coding, 1 or 3 components, sampling factors 1 and 2 in either direction, restart intervals, JFIF (YCbCr) and Adobe
// (RGB / YCbCr) colour conventions.  Chroma is upsampled with the triangle ("fancy") filter every common decoder uses, the inverse
// DCT is evaluated in floating point: decoded bytes agree with libjpeg-turbo to within 3 of 255, 0.35 on average
// (tests/test_abi_and_host.py; integer-IDCT decoders such as stb_image differ from each other by the same amount).
// Lossless, hierarchical and arithmetic-coded files are rejected -- the caller reports the texture as unreadable.
#include "Jpeg.h"

#include <cmath>
#include <cstring>

namespace {

const unsigned char kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {            // canonical code tables of T.81 Annex C / F.2.2.3
    unsigned char vals[256];
    int mincode[17], maxcode[18], valptr[17];
    bool defined = false;
    bool build(const unsigned char* counts, const unsigned char* symbols, int n_symbols) {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k; mincode[len] = code;
            code += counts[len - 1]; k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        if (k != n_symbols || k > 256) return false;
        std::memcpy(vals, symbols, size_t(k));
        defined = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0;
    int pw = 0, ph = 0;                       // plane size, padded to whole MCUs
    std::vector<unsigned char> plane;
    std::vector<short> coef;                  // progressive: 64 coefficients per block (natural order), blocks in plane raster (pw / 8 per row)
};

class Decoder {
public:
    Decoder(const unsigned char* d, size_t n) : d_(d), n_(n) {}
    bool run(int& w, int& h, std::vector<unsigned char>& rgb) {
        if (n_ < 4 || d_[0] != 0xFF || d_[1] != 0xD8) return false;
        pos_ = 2;
        for (;;) {
            int m = next_marker();
            if (m < 0) return false;
            if (m == 0xD9) { if (!progressive_ || !scans_) return false; break; }   // EOI: a progressive file ends here; before any scan it is an error
            if (m == 0xC0 || m == 0xC1) { if (!read_sof()) return false; }
            else if (m == 0xC2) { progressive_ = true; if (!read_sof()) return false; }
            else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) return false;   // lossless / hierarchical / arithmetic
            else if (m == 0xC4) { if (!read_dht()) return false; }
            else if (m == 0xDB) { if (!read_dqt()) return false; }
            else if (m == 0xDD) { if (!read_dri()) return false; }
            else if (m == 0xEE) { if (!read_adobe()) return false; }
            else if (m == 0xDA) {
                if (!progressive_) { if (!read_sos() || !decode_scan()) return false; break; }   // baseline: one interleaved scan
                if (!read_sos_progressive() || !decode_scan_progressive()) return false;
                if (++scans_ > 1000) return false;                     // (a crafted file could go on for ever)
            }
            else if (!skip_segment()) return false;
        }
        if (progressive_ && !finish_progressive()) return false;
        w = width_; h = height_;
        return to_rgb(rgb);
    }

private:
    const unsigned char* d_; size_t n_, pos_ = 0;
    int width_ = 0, height_ = 0, ncomp_ = 0, hmax_ = 1, vmax_ = 1, restart_ = 0;
    bool adobe_ = false; int adobe_transform_ = 0;
    bool progressive_ = false; int scans_ = 0;
    int scan_n_ = 0, scan_comp_[3] = {0, 0, 0}, ss_ = 0, se_ = 0, ah_ = 0, al_ = 0, eobrun_ = 0;   // the current progressive scan
    unsigned short qt_[4][64] = {};
    bool qt_defined_[4] = {false, false, false, false};
    Huffman dc_[4], ac_[4];
    Component comp_[3];
    // entropy-coded segment reader
    unsigned int bitbuf_ = 0; int bitcnt_ = 0; bool hit_marker_ = false;

    int u16(size_t p) const { return (int(d_[p]) << 8) | d_[p + 1]; }
    int next_marker() {
        while (pos_ + 1 < n_) {
            if (d_[pos_] != 0xFF) { pos_++; continue; }
            while (pos_ < n_ && d_[pos_] == 0xFF) pos_++;
            if (pos_ >= n_) return -1;
            const int m = d_[pos_++];
            if (m != 0) return m;
        }
        return -1;
    }
    bool segment(size_t& begin, size_t& end) {
        if (pos_ + 2 > n_) return false;
        const int len = u16(pos_);
        if (len < 2 || pos_ + size_t(len) > n_) return false;
        begin = pos_ + 2; end = pos_ + size_t(len); pos_ = end;
        return true;
    }
    bool skip_segment() { size_t b, e; return segment(b, e); }
    bool read_dri() { size_t b, e; if (!segment(b, e) || e - b < 2) return false; restart_ = u16(b); return true; }
    bool read_adobe() {
        size_t b, e; if (!segment(b, e)) return false;
        if (e - b >= 12 && !std::memcmp(d_ + b, "Adobe", 5)) { adobe_ = true; adobe_transform_ = d_[b + 11]; }
        return true;
    }
    bool read_dqt() {
        size_t b, e; if (!segment(b, e)) return false;
        while (b < e) {
            const int pq = d_[b] >> 4, tq = d_[b] & 15; b++;
            if (tq > 3 || pq > 1 || b + size_t(64 * (pq + 1)) > e) return false;
            for (int i = 0; i < 64; i++) { qt_[tq][kZigzag[i]] = (unsigned short)(pq ? u16(b + 2 * size_t(i)) : d_[b + size_t(i)]); }
            b += size_t(64 * (pq + 1)); qt_defined_[tq] = true;
        }
        return true;
    }
    bool read_dht() {
        size_t b, e; if (!segment(b, e)) return false;
        while (b < e) {
            if (b + 17 > e) return false;
            const int tc = d_[b] >> 4, th = d_[b] & 15;
            if (tc > 1 || th > 3) return false;
            int total = 0; for (int i = 0; i < 16; i++) total += d_[b + 1 + size_t(i)];
            if (total > 256 || b + 17 + size_t(total) > e) return false;
            if (!(tc ? ac_ : dc_)[th].build(d_ + b + 1, d_ + b + 17, total)) return false;
            b += 17 + size_t(total);
        }
        return true;
    }
    bool read_sof() {
        size_t b, e; if (!segment(b, e) || e - b < 6) return false;
        if (d_[b] != 8) return false;                                  // 8-bit samples only
        height_ = u16(b + 1); width_ = u16(b + 3); ncomp_ = d_[b + 5];
        if (width_ <= 0 || height_ <= 0 || (ncomp_ != 1 && ncomp_ != 3) || e - b < size_t(6 + 3 * ncomp_)) return false;
        if (size_t(width_) * size_t(height_) > (size_t(1) << 28)) return false;
        hmax_ = vmax_ = 1;
        for (int i = 0; i < ncomp_; i++) {
            Component& c = comp_[i];
            c.id = d_[b + 6 + 3 * size_t(i)]; c.h = d_[b + 7 + 3 * size_t(i)] >> 4; c.v = d_[b + 7 + 3 * size_t(i)] & 15; c.tq = d_[b + 8 + 3 * size_t(i)];
            if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) return false;
            if (c.h > hmax_) hmax_ = c.h;
            if (c.v > vmax_) vmax_ = c.v;
        }
        if (ncomp_ == 1) { comp_[0].h = comp_[0].v = 1; hmax_ = vmax_ = 1; }      // a single component is never interleaved (A.2.2)
        const int mcux = (width_ + 8 * hmax_ - 1) / (8 * hmax_), mcuy = (height_ + 8 * vmax_ - 1) / (8 * vmax_);
        for (int i = 0; i < ncomp_; i++) {
            Component& c = comp_[i];
            c.pw = mcux * c.h * 8; c.ph = mcuy * c.v * 8;
            c.plane.assign(size_t(c.pw) * size_t(c.ph), 0);
            if (progressive_) c.coef.assign(size_t(c.pw) * size_t(c.ph), 0);
        }
        return true;
    }
    bool read_sos() {
        size_t b, e; if (!segment(b, e) || width_ == 0 || e - b < 1) return false;   // an empty SOS payload has no component count to read
        const int ns = d_[b];
        if (ns != ncomp_ || e - b < size_t(1 + 2 * ns + 3)) return false;   // baseline files carry one interleaved scan
        for (int i = 0; i < ns; i++) {
            const int id = d_[b + 1 + 2 * size_t(i)], t = d_[b + 2 + 2 * size_t(i)];
            int k = -1; for (int j = 0; j < ncomp_; j++) if (comp_[j].id == id) k = j;
            if (k != i) return false;                                  // components in frame order
            comp_[k].td = t >> 4; comp_[k].ta = t & 15;
            if (comp_[k].td > 3 || comp_[k].ta > 3 || !dc_[comp_[k].td].defined || !ac_[comp_[k].ta].defined || !qt_defined_[comp_[k].tq]) return false;
        }
        return true;
    }

    // ---- bit reader over the entropy-coded segment: FF00 -> FF, any other marker ends the segment (zeros are fed after it)
    void fill() {
        while (bitcnt_ <= 24) {
            unsigned int byte = 0;
            if (!hit_marker_ && pos_ < n_) {
                byte = d_[pos_];
                if (byte == 0xFF) {
                    const unsigned int nx = pos_ + 1 < n_ ? d_[pos_ + 1] : 0xD9;
                    if (nx == 0) pos_ += 2;
                    else { hit_marker_ = true; byte = 0; }
                } else pos_++;
            }
            bitbuf_ |= byte << (24 - bitcnt_);
            bitcnt_ += 8;
        }
    }
    int get_bits(int n) {
        if (n == 0) return 0;
        if (bitcnt_ < n) fill();
        const int v = int(bitbuf_ >> (32 - n));
        bitbuf_ <<= n; bitcnt_ -= n;
        return v;
    }
    int decode_symbol(const Huffman& h) {
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (code << 1) | get_bits(1);
            if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + code - h.mincode[len]];
        }
        return -1;
    }
    static int extend(int v, int t) { return (t && v < (1 << (t - 1))) ? v - (1 << t) + 1 : v; }

    bool decode_block(Component& c, float* out) {
        int coef[64]; std::memset(coef, 0, sizeof coef);
        const int t = decode_symbol(dc_[c.td]);
        if (t < 0 || t > 11) return false;
        c.pred += extend(get_bits(t), t);
        if (c.pred < -32768 || c.pred > 32767) return false;           // DC predictor outside the 16-bit range T.81 allows: a crafted / corrupt stream
        coef[0] = c.pred * qt_[c.tq][0];
        for (int k = 1; k < 64;) {
            const int rs = decode_symbol(ac_[c.ta]);
            if (rs < 0) return false;
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) { if (r == 15) { k += 16; continue; } break; }   // ZRL / EOB
            k += r;
            if (k > 63) return false;
            coef[kZigzag[k]] = extend(get_bits(s), s) * qt_[c.tq][kZigzag[k]];
            k++;
        }
        idct(coef, out);
        return true;
    }
    // s(x,y) = 1/4 sum_u sum_v C(u) C(v) S(u,v) cos((2x+1)u pi/16) cos((2y+1)v pi/16)   (T.81 A.3.3), rows then columns
    static void idct(const int* in, float* out) {
        static float basis[8][8]; static bool init = false;
        if (!init) {
            for (int x = 0; x < 8; x++) for (int u = 0; u < 8; u++) basis[x][u] = float((u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0));
            init = true;
        }
        float tmp[64];
        for (int v = 0; v < 8; v++)
            for (int x = 0; x < 8; x++) { float s = 0.f; for (int u = 0; u < 8; u++) s += basis[x][u] * float(in[8 * v + u]); tmp[8 * v + x] = s; }
        for (int x = 0; x < 8; x++)
            for (int y = 0; y < 8; y++) { float s = 0.f; for (int v = 0; v < 8; v++) s += basis[y][v] * tmp[8 * v + x]; out[8 * y + x] = s; }
    }
    void restart() {
        bitbuf_ = 0; bitcnt_ = 0; hit_marker_ = false;
        for (size_t p = pos_; p + 1 < n_ && p < pos_ + 8; p++)          // the RSTn marker follows the padded last byte of the interval
            if (d_[p] == 0xFF && d_[p + 1] >= 0xD0 && d_[p + 1] <= 0xD7) { pos_ = p + 2; break; }
        for (int i = 0; i < ncomp_; i++) comp_[i].pred = 0;
    }
    bool decode_scan() {
        const int mcux = (width_ + 8 * hmax_ - 1) / (8 * hmax_), mcuy = (height_ + 8 * vmax_ - 1) / (8 * vmax_);
        bitbuf_ = 0; bitcnt_ = 0; hit_marker_ = false;
        for (int i = 0; i < ncomp_; i++) comp_[i].pred = 0;
        int until_restart = restart_;
        float px[64];
        for (int my = 0; my < mcuy; my++)
            for (int mx = 0; mx < mcux; mx++) {
                if (restart_ && until_restart == 0) { restart(); until_restart = restart_; }
                for (int i = 0; i < ncomp_; i++) {
                    Component& c = comp_[i];
                    for (int by = 0; by < c.v; by++)
                        for (int bx = 0; bx < c.h; bx++) {
                            if (!decode_block(c, px)) return false;
                            unsigned char* dst = &c.plane[size_t((my * c.v + by) * 8) * size_t(c.pw) + size_t((mx * c.h + bx) * 8)];
                            for (int y = 0; y < 8; y++)
                                for (int x = 0; x < 8; x++) {
                                    const float s = px[8 * y + x] + 128.0f;
                                    dst[size_t(y) * size_t(c.pw) + size_t(x)] = (unsigned char)(s <= 0.f ? 0 : (s >= 255.f ? 255 : int(s + 0.5f)));
                                }
                        }
                }
                if (restart_) until_restart--;
            }
        return true;
    }


    // ---- progressive mode (T.81 Annex G)
    bool read_sos_progressive() {
        size_t b, e; if (!segment(b, e) || width_ == 0 || e - b < 1) return false;
        const int ns = d_[b];
        if (ns < 1 || ns > ncomp_ || e - b < size_t(1 + 2 * ns + 3)) return false;
        scan_n_ = ns;
        for (int i = 0; i < ns; i++) {
            const int id = d_[b + 1 + 2 * size_t(i)], t = d_[b + 2 + 2 * size_t(i)];
            int k = -1; for (int j = 0; j < ncomp_; j++) if (comp_[j].id == id) k = j;
            if (k < 0 || (i && k <= scan_comp_[i - 1])) return false;     // known components, in frame order
            scan_comp_[i] = k; comp_[k].td = t >> 4; comp_[k].ta = t & 15;
            if (comp_[k].td > 3 || comp_[k].ta > 3) return false;
        }
        ss_ = d_[b + 1 + 2 * size_t(ns)]; se_ = d_[b + 2 + 2 * size_t(ns)]; ah_ = d_[b + 3 + 2 * size_t(ns)] >> 4; al_ = d_[b + 3 + 2 * size_t(ns)] & 15;
        if (ss_ > se_ || se_ > 63 || al_ > 13 || ah_ > 13 || (ss_ == 0 && se_ != 0) || (ss_ > 0 && ns != 1)) return false;   // DC scans carry DC only; AC scans one component
        for (int i = 0; i < ns; i++) {
            const Component& c = comp_[scan_comp_[i]];
            if (ss_ == 0 ? (ah_ == 0 && !dc_[c.td].defined) : !ac_[c.ta].defined) return false;
        }
        return true;
    }
    int get_bit() { return get_bits(1); }
    bool dc_first(Component& c, short* blk) {
        const int t = decode_symbol(dc_[c.td]);
        if (t < 0 || t > 11) return false;
        c.pred += extend(get_bits(t), t);
        if (c.pred < -32768 || c.pred > 32767) return false;
        blk[0] = short(c.pred * (1 << al_));
        return true;
    }
    void dc_refine(short* blk) { if (get_bit()) blk[0] = short(blk[0] | (1 << al_)); }
    bool ac_first(const Component& c, short* blk) {
        if (eobrun_ > 0) { eobrun_--; return true; }
        for (int k = ss_; k <= se_;) {
            const int rs = decode_symbol(ac_[c.ta]);
            if (rs < 0) return false;
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (r < 15) { eobrun_ = (1 << r) - 1; if (r) eobrun_ += get_bits(r); break; }   // EOBn: this block and eobrun_ more end here
                k += 16; continue;                                                               // ZRL
            }
            k += r;
            if (k > se_) return false;
            blk[kZigzag[k]] = short(extend(get_bits(sz), sz) * (1 << al_));
            k++;
        }
        return true;
    }
    // G.1.2.3: a refinement scan sends one more bit of every coefficient that is already non-zero (a correction bit) and the newly
    // non-zero ones (magnitude 1 << al_) in between, with zero runs that count only coefficients that are still zero
    void correct(short& v) { if (get_bit() && (v & (1 << al_)) == 0) v = short(v > 0 ? v + (1 << al_) : v - (1 << al_)); }
    bool ac_refine(const Component& c, short* blk) {
        int k = ss_;
        if (eobrun_ > 0) { eobrun_--; for (; k <= se_; k++) { short& v = blk[kZigzag[k]]; if (v != 0) correct(v); } return true; }
        while (k <= se_) {
            const int rs = decode_symbol(ac_[c.ta]);
            if (rs < 0) return false;
            int r = rs >> 4; const int sz = rs & 15; int nv = 0;
            if (sz == 0) {
                if (r < 15) { eobrun_ = (1 << r) - 1; if (r) eobrun_ += get_bits(r); r = 64; }   // end of band for this block: corrections for the rest of it
            } else {
                if (sz != 1) return false;
                nv = get_bit() ? (1 << al_) : -(1 << al_);
            }
            while (k <= se_) {
                short& v = blk[kZigzag[k++]];
                if (v != 0) correct(v);
                else { if (r == 0) { if (nv) v = short(nv); break; } r--; }
            }
        }
        return true;
    }
    bool decode_scan_progressive() {
        bitbuf_ = 0; bitcnt_ = 0; hit_marker_ = false; eobrun_ = 0;
        for (int i = 0; i < ncomp_; i++) comp_[i].pred = 0;
        int until_restart = restart_;
        auto block_of = [](Component& c, int bx, int by) { return &c.coef[(size_t(by) * size_t(c.pw / 8) + size_t(bx)) * 64]; };
        auto one = [&](Component& c, short* blk) -> bool {
            if (ss_ == 0) { if (ah_ == 0) return dc_first(c, blk); dc_refine(blk); return true; }
            return ah_ == 0 ? ac_first(c, blk) : ac_refine(c, blk);
        };
        auto at_restart = [&]() { if (restart_ && until_restart == 0) { restart(); eobrun_ = 0; until_restart = restart_; } };
        if (scan_n_ == 1) {                                               // non-interleaved: the component's own blocks in raster order (A.2.3)
            Component& c = comp_[scan_comp_[0]];
            const int cw = (width_ * c.h + hmax_ - 1) / hmax_, chh = (height_ * c.v + vmax_ - 1) / vmax_;
            const int bw = (cw + 7) / 8, bh = (chh + 7) / 8;
            for (int by = 0; by < bh; by++)
                for (int bx = 0; bx < bw; bx++) {
                    at_restart();
                    if (!one(c, block_of(c, bx, by))) return false;
                    if (restart_) until_restart--;
                }
        } else {                                                          // interleaved (DC scans only): MCU order
            const int mcux = (width_ + 8 * hmax_ - 1) / (8 * hmax_), mcuy = (height_ + 8 * vmax_ - 1) / (8 * vmax_);
            for (int my = 0; my < mcuy; my++)
                for (int mx = 0; mx < mcux; mx++) {
                    at_restart();
                    for (int i = 0; i < scan_n_; i++) {
                        Component& c = comp_[scan_comp_[i]];
                        for (int by = 0; by < c.v; by++)
                            for (int bx = 0; bx < c.h; bx++)
                                if (!one(c, block_of(c, mx * c.h + bx, my * c.v + by))) return false;
                    }
                    if (restart_) until_restart--;
                }
        }
        return true;
    }
    bool finish_progressive() {                                           // dequantise + inverse DCT of every block, once all scans are in
        float px[64]; int coef[64];
        for (int i = 0; i < ncomp_; i++) {
            Component& c = comp_[i];
            if (!qt_defined_[c.tq]) return false;
            const int bw = c.pw / 8, bh = c.ph / 8;
            for (int by = 0; by < bh; by++)
                for (int bx = 0; bx < bw; bx++) {
                    const short* blk = &c.coef[(size_t(by) * size_t(bw) + size_t(bx)) * 64];
                    for (int k = 0; k < 64; k++) coef[k] = int(blk[k]) * qt_[c.tq][k];
                    idct(coef, px);
                    unsigned char* dst = &c.plane[size_t(by * 8) * size_t(c.pw) + size_t(bx * 8)];
                    for (int y = 0; y < 8; y++)
                        for (int x = 0; x < 8; x++) {
                            const float v = px[8 * y + x] + 128.0f;
                            dst[size_t(y) * size_t(c.pw) + size_t(x)] = (unsigned char)(v <= 0.f ? 0 : (v >= 255.f ? 255 : int(v + 0.5f)));
                        }
                }
        }
        return true;
    }

    // One output row of a component at full resolution.  Subsampled components use the triangle filter: every output sample
    // is 3/4 of the nearer and 1/4 of the farther input sample, separably in both directions, edges replicated.
    void full_res_row(const Component& c, int y, std::vector<int>& row16) const {      // values scaled by 16
        const int sx = hmax_ / c.h, sy = vmax_ / c.v;
        const int cw = (width_ + sx - 1) / sx, chh = (height_ + sy - 1) / sy;
        std::vector<int> v4(size_t(cw) + 2);                             // vertical pass, scaled by 4
        if (sy == 1) { const unsigned char* r = &c.plane[size_t(y) * size_t(c.pw)]; for (int x = 0; x < cw; x++) v4[size_t(x)] = 4 * r[x]; }
        else {
            const int yn = y >> 1; int yf = (y & 1) ? yn + 1 : yn - 1;
            if (yf < 0) yf = 0;
            if (yf >= chh) yf = chh - 1;
            const unsigned char* rn = &c.plane[size_t(yn) * size_t(c.pw)]; const unsigned char* rf = &c.plane[size_t(yf) * size_t(c.pw)];
            for (int x = 0; x < cw; x++) v4[size_t(x)] = 3 * rn[x] + rf[x];
        }
        row16.resize(size_t(width_));
        if (sx == 1) { for (int x = 0; x < width_; x++) row16[size_t(x)] = 4 * v4[size_t(x)]; }
        else
            for (int x = 0; x < width_; x++) {
                const int xn = x >> 1; int xf = (x & 1) ? xn + 1 : xn - 1;
                if (xf < 0) xf = 0;
                if (xf >= cw) xf = cw - 1;
                row16[size_t(x)] = 3 * v4[size_t(xn)] + v4[size_t(xf)];
            }
    }
    bool to_rgb(std::vector<unsigned char>& rgb) const {
        rgb.resize(size_t(width_) * size_t(height_) * 3);
        auto clamp8 = [](float v) { return (unsigned char)(v <= 0.f ? 0 : (v >= 255.f ? 255 : int(v + 0.5f))); };
        std::vector<int> r0, r1, r2;
        const bool ycc = ncomp_ == 3 && !(adobe_ && adobe_transform_ == 0);
        for (int y = 0; y < height_; y++) {
            full_res_row(comp_[0], y, r0);
            if (ncomp_ == 3) { full_res_row(comp_[1], y, r1); full_res_row(comp_[2], y, r2); }
            unsigned char* o = &rgb[size_t(y) * size_t(width_) * 3];
            for (int x = 0; x < width_; x++) {
                const float a = float(r0[size_t(x)]) * (1.0f / 16.0f);
                if (ncomp_ == 1) { o[3 * x] = o[3 * x + 1] = o[3 * x + 2] = clamp8(a); continue; }
                const float b = float(r1[size_t(x)]) * (1.0f / 16.0f), c = float(r2[size_t(x)]) * (1.0f / 16.0f);
                if (!ycc) { o[3 * x] = clamp8(a); o[3 * x + 1] = clamp8(b); o[3 * x + 2] = clamp8(c); continue; }
                const float cb = b - 128.0f, cr = c - 128.0f;                // JFIF: ITU-R BT.601 full range
                o[3 * x] = clamp8(a + 1.402f * cr);
                o[3 * x + 1] = clamp8(a - 0.344136f * cb - 0.714136f * cr);
                o[3 * x + 2] = clamp8(a + 1.772f * cb);
            }
        }
        return true;
    }
};

}  // namespace

bool load_jpeg(const std::vector<unsigned char>& file, int& w, int& h, std::vector<unsigned char>& rgb) {
    Decoder d(file.data(), file.size());
    return d.run(w, h, rgb);
}
