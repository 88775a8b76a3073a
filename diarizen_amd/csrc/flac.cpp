// flac.cpp — native FLAC decoder of the audio-ingest row: the reference loads "anything torchaudio can"
// (diarizen/pipelines/inference.py:127 -> torchaudio.load), and FLAC is what meeting corpora ship besides WAV; torchaudio /
// libFLAC / libsndfile are not in this image, so the format is decoded here from its published specification (RFC 9639 /
// xiph.org "FLAC format"): STREAMINFO, frame headers (fixed and variable block size), CONSTANT / VERBATIM / FIXED (order 0-4) /
// LPC (order 1-32) subframes, Rice-coded residuals (4- and 5-bit parameters, escaped partitions), wasted bits, the three
// stereo decorrelations, 4-32 bits per sample.  Host code (no device work): one pass over the bit stream, ~100 M samples/s.
//
// Integrity is checked, not assumed: every frame header's CRC-8 and every frame's CRC-16 are verified while decoding, and
// diarizen_amd/audio.py verifies the STREAMINFO MD5 of the decoded PCM — a file whose MD5 field is set cannot decode to wrong
// samples silently.
//
// C ABI (include/dzn.h):  dzn_flac_info  -> stream parameters;  dzn_flac_decode -> int32 samples, interleaved [samples][channels].
#include <stdint.h>

#include <vector>
#include <string.h>

#include "../../include/dzn.h"

namespace {

struct BitReader {
  const uint8_t* p;
  size_t n, pos;        // pos = byte position
  uint64_t acc;         // bit accumulator (MSB first)
  int bits;             // valid bits in acc
  bool bad;
  BitReader(const uint8_t* d, size_t len, size_t at) : p(d), n(len), pos(at), acc(0), bits(0), bad(false) {}
  inline void fill() {
    while (bits <= 56 && pos < n) {
      acc |= (uint64_t)p[pos++] << (56 - bits);
      bits += 8;
    }
  }
  inline uint32_t get(int k) {          // k in [0, 32]
    if (k == 0) return 0;
    if (bits < k) {
      fill();
      if (bits < k) { bad = true; return 0; }
    }
    const uint32_t v = (uint32_t)(acc >> (64 - k));
    acc <<= k;                                        // k <= 32
    bits -= k;
    return v;
  }
  inline int32_t get_signed(int k) {
    if (k == 0) return 0;
    const uint32_t v = get(k);
    const uint32_t m = 1u << (k - 1);
    return (int32_t)((v ^ m) - m);
  }
  inline int64_t get_signed64(int k) {   // k up to 33 (side channel of 32-bit audio)
    if (k <= 32) return get_signed(k);
    const uint64_t hi = get(k - 32), lo = get(32);
    const uint64_t v = (hi << 32) | lo;
    const uint64_t m = 1ull << (k - 1);
    return (int64_t)((v ^ m) - m);
  }
  inline uint32_t unary() {              // number of 0 bits before the next 1 bit
    uint32_t q = 0;
    for (;;) {
      if (bits == 0) {
        fill();
        if (bits == 0) { bad = true; return 0; }
      }
      if (acc == 0) {                    // all valid bits are zero
        q += bits;
        bits = 0;
        continue;
      }
      const int lz = __builtin_clzll(acc);
      if (lz >= bits) {                  // cannot happen when acc != 0 within `bits`, kept for safety
        q += bits;
        acc = 0;
        bits = 0;
        continue;
      }
      q += lz;
      acc = lz == 63 ? 0 : acc << (lz + 1);         // (a shift by 64 is undefined)
      bits -= (lz + 1);
      return q;
    }
  }
  inline void align() {                  // drop to the next byte boundary
    const int r = bits & 7;
    acc <<= r;
    bits -= r;
  }
  inline size_t byte_pos() const { return pos - (size_t)(bits >> 3); }   // only meaningful when aligned
};

uint8_t crc8(const uint8_t* d, size_t n) {
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= d[i];
    for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? ((c << 1) ^ 0x07) : (c << 1));
  }
  return c;
}

struct Crc16Table {
  uint16_t t[256];
  Crc16Table() {
    for (int i = 0; i < 256; ++i) {
      uint16_t c = (uint16_t)(i << 8);
      for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1));
      t[i] = c;
    }
  }
};

uint16_t crc16(const uint8_t* d, size_t n) {
  // function-local static: initialised exactly once under the language's guard - the loader thread of diarize_many and a
  // caller's own load_flac may decode at the same time (ctypes releases the GIL); a hand-rolled `static bool init` could let
  // the second thread read a half-written table and report a good frame as corrupt
  static const Crc16Table table_obj;
  const uint16_t* table = table_obj.t;
  uint16_t c = 0;
  for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ table[(c >> 8) ^ d[i]]);
  return c;
}

struct StreamInfo {
  int min_block, max_block, sample_rate, channels, bps;
  int64_t total;
  uint8_t md5[16];
  size_t first_frame;
};

// "fLaC" + metadata blocks; STREAMINFO must come first
int parse_header(const uint8_t* d, size_t n, StreamInfo& si) {
  if (n < 42 || memcmp(d, "fLaC", 4) != 0) return DZN_E_INVALID;
  size_t at = 4;
  bool have = false;
  for (;;) {
    if (at + 4 > n) return DZN_E_INVALID;
    const bool last = (d[at] & 0x80) != 0;
    const int type = d[at] & 0x7f;
    const size_t len = ((size_t)d[at + 1] << 16) | ((size_t)d[at + 2] << 8) | d[at + 3];
    at += 4;
    if (at + len > n) return DZN_E_INVALID;
    if (type == 0) {
      if (len < 34) return DZN_E_INVALID;
      const uint8_t* s = d + at;
      si.min_block = (s[0] << 8) | s[1];
      si.max_block = (s[2] << 8) | s[3];
      si.sample_rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
      si.channels = ((s[12] >> 1) & 7) + 1;
      si.bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
      si.total = ((int64_t)(s[13] & 0x0f) << 32) | ((int64_t)s[14] << 24) | ((int64_t)s[15] << 16) | ((int64_t)s[16] << 8) | s[17];
      memcpy(si.md5, s + 18, 16);
      have = true;
    }
    at += len;
    if (last) break;
  }
  if (!have || si.sample_rate <= 0 || si.bps < 4 || si.bps > 32) return DZN_E_INVALID;
  si.first_frame = at;
  return DZN_OK;
}

// residual of one subframe into out[pred_order .. blocksize)
bool read_residual(BitReader& br, int64_t* out, int blocksize, int pred_order) {
  const int method = (int)br.get(2);
  if (method > 1) return false;
  const int pbits = method == 0 ? 4 : 5;
  const uint32_t esc = method == 0 ? 15u : 31u;
  const int porder = (int)br.get(4);
  const int nparts = 1 << porder;
  if ((blocksize >> porder) << porder != blocksize && porder > 0) return false;
  int i = pred_order;
  for (int part = 0; part < nparts; ++part) {
    int cnt = porder == 0 ? blocksize - pred_order : (part == 0 ? (blocksize >> porder) - pred_order : (blocksize >> porder));
    if (cnt < 0 || i + cnt > blocksize) return false;
    const uint32_t k = br.get(pbits);
    if (k == esc) {
      const int nb = (int)br.get(5);
      for (int j = 0; j < cnt; ++j) out[i++] = br.get_signed(nb);
    } else {
      for (int j = 0; j < cnt; ++j) {
        const uint32_t q = br.unary();
        const uint64_t v = ((uint64_t)q << k) | br.get((int)k);
        out[i++] = (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
      }
    }
    if (br.bad) return false;
  }
  return i == blocksize;
}

bool read_subframe(BitReader& br, int64_t* s, int blocksize, int bps) {
  if (br.get(1) != 0) return false;
  const int type = (int)br.get(6);
  int wasted = 0;
  if (br.get(1)) wasted = (int)br.unary() + 1;
  if (br.bad || wasted >= bps) return false;
  bps -= wasted;
  if (type == 0) {                                    // CONSTANT
    const int64_t v = br.get_signed64(bps);
    for (int i = 0; i < blocksize; ++i) s[i] = v;
  } else if (type == 1) {                             // VERBATIM
    for (int i = 0; i < blocksize; ++i) s[i] = br.get_signed64(bps);
  } else if (type >= 8 && type <= 12) {               // FIXED, order type - 8
    const int order = type - 8;
    if (order > blocksize) return false;
    for (int i = 0; i < order; ++i) s[i] = br.get_signed64(bps);
    if (!read_residual(br, s, blocksize, order)) return false;
    for (int i = order; i < blocksize; ++i) {
      switch (order) {
        case 0: break;
        case 1: s[i] += s[i - 1]; break;
        case 2: s[i] += 2 * s[i - 1] - s[i - 2]; break;
        case 3: s[i] += 3 * s[i - 1] - 3 * s[i - 2] + s[i - 3]; break;
        default: s[i] += 4 * s[i - 1] - 6 * s[i - 2] + 4 * s[i - 3] - s[i - 4]; break;
      }
    }
  } else if (type >= 32) {                            // LPC, order type - 31
    const int order = type - 31;
    if (order > blocksize) return false;
    for (int i = 0; i < order; ++i) s[i] = br.get_signed64(bps);
    const int prec = (int)br.get(4) + 1;
    if (prec == 16) return false;
    const int shift = br.get_signed(5);
    if (shift < 0) return false;
    int32_t coef[32];
    for (int j = 0; j < order; ++j) coef[j] = br.get_signed(prec);
    if (!read_residual(br, s, blocksize, order)) return false;
    for (int i = order; i < blocksize; ++i) {
      int64_t acc = 0;
      for (int j = 0; j < order; ++j) acc += (int64_t)coef[j] * s[i - 1 - j];
      s[i] += acc >> shift;
    }
  } else {
    return false;                                     // reserved subframe types
  }
  if (br.bad) return false;
  if (wasted)
    for (int i = 0; i < blocksize; ++i) s[i] = (int64_t)((uint64_t)s[i] << wasted);
  return true;
}

}  // namespace

extern "C" int dzn_flac_info(const uint8_t* data, size_t n, int32_t* sample_rate, int32_t* channels, int32_t* bits_per_sample,
                             int64_t* total_samples, uint8_t* md5_16) {
  if (!data) return DZN_E_INVALID;
  StreamInfo si{};
  const int rc = parse_header(data, n, si);
  if (rc != DZN_OK) return rc;
  if (sample_rate) *sample_rate = si.sample_rate;
  if (channels) *channels = si.channels;
  if (bits_per_sample) *bits_per_sample = si.bps;
  if (total_samples) *total_samples = si.total;
  if (md5_16) memcpy(md5_16, si.md5, 16);
  return DZN_OK;
}

// out = int32 [capacity_samples][channels] interleaved; *decoded = samples (per channel) written.  Streams whose STREAMINFO
// carries no total (0: unknown length) decode until the data end; capacity must then be large enough (DZN_E_NOMEM otherwise).
extern "C" int dzn_flac_decode(const uint8_t* data, size_t n, int32_t* out, int64_t capacity_samples, int64_t* decoded) {
  if (!data || !out || capacity_samples < 0) return DZN_E_INVALID;
  StreamInfo si{};
  int rc = parse_header(data, n, si);
  if (rc != DZN_OK) return rc;
  const int C = si.channels;
  int64_t done = 0;
  size_t at = si.first_frame;
  // per-call scratch sized by the stream's own maximum block (STREAMINFO; 65535 when it states none) instead of 4 MiB of
  // thread-local storage paid by every thread that ever decodes
  const int max_bs = si.max_block >= 16 && si.max_block <= 65535 ? si.max_block : 65536;
  std::vector<int64_t> scratch;
  try {
    scratch.resize((size_t)C * (size_t)max_bs);
  } catch (...) {
    return DZN_E_NOMEM;
  }
  int64_t* buf[8];
  for (int c = 0; c < 8; ++c) buf[c] = scratch.data() + (size_t)(c < C ? c : 0) * (size_t)max_bs;
  while (at + 2 <= n && (si.total == 0 || done < si.total)) {
    if (!(data[at] == 0xff && (data[at + 1] & 0xfe) == 0xf8)) {
      if (si.total == 0) break;                       // trailing bytes behind the last frame of a stream of unknown length
      return DZN_E_INVALID;
    }
    BitReader br(data, n, at);
    br.get(14);
    if (br.get(1)) return DZN_E_INVALID;
    br.get(1);                                        // blocking strategy: the coded number is a frame or a sample number — unused
    const int bs_code = (int)br.get(4), sr_code = (int)br.get(4), ch_code = (int)br.get(4), ss_code = (int)br.get(3);
    if (br.get(1)) return DZN_E_INVALID;
    {                                                 // UTF-8-like coded number
      uint32_t b0 = br.get(8);
      int extra = 0;
      if (b0 & 0x80) {
        while (b0 & (0x80u >> extra)) ++extra;
        if (extra < 2 || extra > 7) return DZN_E_INVALID;
        extra -= 1;
      }
      for (int i = 0; i < extra; ++i)
        if ((br.get(8) & 0xc0) != 0x80) return DZN_E_INVALID;
    }
    int blocksize;
    if (bs_code == 0) return DZN_E_INVALID;
    else if (bs_code == 1) blocksize = 192;
    else if (bs_code <= 5) blocksize = 576 << (bs_code - 2);
    else if (bs_code == 6) blocksize = (int)br.get(8) + 1;
    else if (bs_code == 7) blocksize = (int)br.get(16) + 1;
    else blocksize = 256 << (bs_code - 8);
    if (sr_code == 12) br.get(8);
    else if (sr_code == 13 || sr_code == 14) br.get(16);
    else if (sr_code == 15) return DZN_E_INVALID;
    static const int ss_table[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    int bps = ss_table[ss_code];
    if (bps < 0) return DZN_E_INVALID;
    if (bps == 0) bps = si.bps;
    if (br.bad) return DZN_E_INVALID;
    const size_t hdr_end = br.byte_pos();
    const uint8_t want8 = (uint8_t)br.get(8);
    if (crc8(data + at, hdr_end - at) != want8) return DZN_E_INVALID;
    int nch;
    if (ch_code < 8) nch = ch_code + 1;
    else if (ch_code <= 10) nch = 2;
    else return DZN_E_INVALID;
    if (nch != C || blocksize > max_bs || blocksize < 1) return DZN_E_INVALID;
    for (int c = 0; c < nch; ++c) {
      int sub_bps = bps;
      if ((ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1)) sub_bps += 1;   // the side channel
      if (!read_subframe(br, buf[c], blocksize, sub_bps)) return DZN_E_INVALID;
    }
    br.align();
    const size_t body_end = br.byte_pos();
    const uint16_t want16 = (uint16_t)br.get(16);
    if (br.bad || crc16(data + at, body_end - at) != want16) return DZN_E_INVALID;
    if (ch_code == 8) {                               // left, side
      for (int i = 0; i < blocksize; ++i) buf[1][i] = buf[0][i] - buf[1][i];
    } else if (ch_code == 9) {                        // side, right
      for (int i = 0; i < blocksize; ++i) buf[0][i] = buf[0][i] + buf[1][i];
    } else if (ch_code == 10) {                       // mid, side
      for (int i = 0; i < blocksize; ++i) {
        const int64_t side = buf[1][i];
        const int64_t mid = (int64_t)((uint64_t)buf[0][i] << 1) | (side & 1);
        buf[0][i] = (mid + side) >> 1;
        buf[1][i] = (mid - side) >> 1;
      }
    }
    int64_t take = blocksize;
    if (si.total > 0 && done + take > si.total) take = si.total - done;
    if (done + take > capacity_samples) return DZN_E_NOMEM;
    for (int c = 0; c < C; ++c) {
      int32_t* o = out + done * C + c;
      for (int64_t i = 0; i < take; ++i) o[i * C] = (int32_t)buf[c][i];
    }
    done += take;
    at = body_end + 2;
  }
  if (si.total > 0 && done != si.total) return DZN_E_INVALID;
  if (decoded) *decoded = done;
  return DZN_OK;
}
