// tables.h - host-side construction of vdl2::Tables (read-only lookup data for
// the walker and the burst decoder).  Everything is derived, not copied:
//   preamble phases / regression abscissae   src/demod.c:84-96,107-124
//   header (25,20) code: coset-leader table rebuilt from the parity-check matrix H
//                                            src/decode.c:55-100
//   GF(2^8) log/antilog for poly 0x187       src/rs.c:28, src/libfec/init_rs.h:52-66
//   descrambler PRBS x^15+x+1 from 0x6959    src/bitstream.c:94-107, src/decode.c:50
//   FCS table (CRC-16-CCITT, reflected)      src/crc.c:21-64
#pragma once
#include <cstring>
#include "vdl2_core.h"

namespace vdl2 {

inline uint32_t hdr_syndrome_host(uint32_t w, const uint32_t H[kHdrParBits]) {
	uint32_t s = 0;
	for(int i = 0; i < kHdrParBits; i++) s |= parity32(w & H[i]) << (kHdrParBits - 1 - i);
	return s;
}

inline void build_tables(Tables &T) {
	std::memset(&T, 0, sizeof T);
	// cumulative unique-word phases, in units of pi/4, wrapped to (-pi, pi] as the reference lists them
	static const int q[kPreamble] = { 0, 3, -3, 1, 1, 2, 0, 4, -3, 4, -2, 3, 1, -2, -3, 0 };
	for(int i = 0; i < kPreamble; i++) T.pr_phase[i] = (float)(q[i] * M_PI / 4);
	float mean_x = 0.f;
	for(int i = 0; i < kPreamble; i++) mean_x += i;
	mean_x /= kPreamble;
	T.lr_den = 0.f;
	for(int i = 0; i < kPreamble; i++) {
		T.lrx[i] = i - mean_x;
		T.lr_den += (i - mean_x) * (i - mean_x);
	}
	static const uint8_t gray[8] = { 0, 1, 3, 2, 6, 7, 5, 4 };
	std::memcpy(T.gray, gray, 8);

	// H rows: columns 24..5 = message part, columns 4..0 = identity
	static const uint32_t H[kHdrParBits] = { 0x001FFF0u, 0x07E1FE8u, 0x18E61E4u, 0x1B6A662u, 0x0D3CAA1u };
	std::memcpy(T.hdr_H, H, sizeof H);
	for(int bit = 0; bit < kHdrBits; bit++) {
		uint32_t e = 1u << bit, s = hdr_syndrome_host(e, H);
		T.hdr_fix[s] = e; T.hdr_weight[s] = 1;
	}
	// the six syndromes no single-bit error produces are mapped to these double-bit patterns
	static const uint8_t pairs[6][2] = { {23, 2}, {23, 1}, {24, 20}, {23, 14}, {23, 15}, {24, 16} };
	for(int i = 0; i < 6; i++) {
		uint32_t e = (1u << pairs[i][0]) | (1u << pairs[i][1]), s = hdr_syndrome_host(e, H);
		T.hdr_fix[s] = e; T.hdr_weight[s] = 2;
	}

	int sr = 1;
	T.gf_log[0] = 255;
	for(int i = 0; i < 255; i++) {
		T.gf_log[sr] = (uint8_t)i; T.gf_exp[i] = (uint8_t)sr;
		sr <<= 1;
		if(sr & 0x100) sr ^= 0x187;
		sr &= 255;
	}
	for(int i = 255; i < 512; i++) T.gf_exp[i] = T.gf_exp[i - 255];

	// reflected CRC-16-CCITT: one table step = eight shifts with the reversed polynomial 0x8408
	for(uint32_t i = 0; i < 256; i++) {
		uint32_t c = i;
		for(int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ 0x8408u : c >> 1;
		T.crc16[i] = (uint16_t)c;
	}

	uint32_t l = kLfsrIv;
	for(int i = 0; i < kPrbsBits; i++) {
		uint32_t bit = (l ^ (l >> 14)) & 1u;
		l = (l >> 1) | (bit << 14);
		T.prbs[i] = (uint8_t)bit;
	}
	// the burst decoder takes the scrambling sequence an octet at a time (bitstream.c:70-81 packs LSB first)
	for(int i = 0; i < kMaxOctets; i++) {
		uint32_t o = 0;
		for(int j = 0; j < 8; j++) { const int bb = kHdrBits + 8 * i + j; if(bb < kPrbsBits) o |= (uint32_t)T.prbs[bb] << j; }
		T.prbs_oct[i] = (uint8_t)o;
	}
}

}  // namespace vdl2
