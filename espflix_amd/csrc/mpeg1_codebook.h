// mpeg1_codebook.h -- ISO/IEC 11172-2 annex B variable-length code books in numeric form.
//
// One entry per code word: the code's bits right-aligned in `code`, its length, and the value
// it stands for.  The decoder builds flat look-up tables from these at context creation
// (build_parse_tables, efx_tables.cpp); the synthetic-stream encoder (gen/efx_gen.cpp) uses them to emit bits.
// The reference carries the same books as packed binary-tree tables (player.cpp:59-116) and
// as a hand-unrolled prefix decoder for the DCT coefficients (player.cpp:532-644);
// tests/test_codebook_vs_reference.py walks those and compares them with this file.
#pragma once
#include <cstdint>

namespace efx {

struct VlcCode {
    uint16_t code;  // right-aligned code bits
    uint8_t len;    // code length in bits
    int16_t value;
};

struct DctCode {
    uint16_t code;
    uint8_t len;
    uint8_t run;
    uint8_t level;  // magnitude; the sign bit follows the code
};

// table B-1, macroblock_address_increment; 34 = macroblock_stuffing, 35 = macroblock_escape
static const VlcCode kMbaCodes[35] = {
    {0x001, 1, 1}, {0x003, 3, 2}, {0x002, 3, 3}, {0x003, 4, 4}, {0x002, 4, 5}, {0x003, 5, 6},
    {0x002, 5, 7}, {0x007, 7, 8}, {0x006, 7, 9}, {0x00B, 8, 10}, {0x00A, 8, 11}, {0x009, 8, 12},
    {0x008, 8, 13}, {0x007, 8, 14}, {0x006, 8, 15}, {0x017, 10, 16}, {0x016, 10, 17},
    {0x015, 10, 18}, {0x014, 10, 19}, {0x013, 10, 20}, {0x012, 10, 21}, {0x023, 11, 22},
    {0x022, 11, 23}, {0x021, 11, 24}, {0x020, 11, 25}, {0x01F, 11, 26}, {0x01E, 11, 27},
    {0x01D, 11, 28}, {0x01C, 11, 29}, {0x01B, 11, 30}, {0x01A, 11, 31}, {0x019, 11, 32},
    {0x018, 11, 33}, {0x00F, 11, 34}, {0x008, 11, 35},
};

// table B-2a, macroblock_type in I pictures; bit0 intra, bit4 quant
static const VlcCode kTypeICodes[2] = {
    {0x001, 1, 1}, {0x001, 2, 17},
};

// table B-2b, macroblock_type in P pictures; bit0 intra, bit1 pattern, bit3 motion forward, bit4 quant
static const VlcCode kTypePCodes[7] = {
    {0x001, 1, 10}, {0x001, 2, 2}, {0x001, 3, 8}, {0x003, 5, 1}, {0x002, 5, 26}, {0x001, 5, 18},
    {0x001, 6, 17},
};

// table B-3, coded_block_pattern
static const VlcCode kCbpCodes[63] = {
    {0x007, 3, 60}, {0x00D, 4, 4}, {0x00C, 4, 8}, {0x00B, 4, 16}, {0x00A, 4, 32}, {0x013, 5, 12},
    {0x012, 5, 48}, {0x011, 5, 20}, {0x010, 5, 40}, {0x00F, 5, 28}, {0x00E, 5, 44}, {0x00D, 5, 52},
    {0x00C, 5, 56}, {0x00B, 5, 1}, {0x00A, 5, 61}, {0x009, 5, 2}, {0x008, 5, 62}, {0x00F, 6, 24},
    {0x00E, 6, 36}, {0x00D, 6, 3}, {0x00C, 6, 63}, {0x017, 7, 5}, {0x016, 7, 9}, {0x015, 7, 17},
    {0x014, 7, 33}, {0x013, 7, 6}, {0x012, 7, 10}, {0x011, 7, 18}, {0x010, 7, 34}, {0x01F, 8, 7},
    {0x01E, 8, 11}, {0x01D, 8, 19}, {0x01C, 8, 35}, {0x01B, 8, 13}, {0x01A, 8, 49}, {0x019, 8, 21},
    {0x018, 8, 41}, {0x017, 8, 14}, {0x016, 8, 50}, {0x015, 8, 22}, {0x014, 8, 42}, {0x013, 8, 15},
    {0x012, 8, 51}, {0x011, 8, 23}, {0x010, 8, 43}, {0x00F, 8, 25}, {0x00E, 8, 37}, {0x00D, 8, 26},
    {0x00C, 8, 38}, {0x00B, 8, 29}, {0x00A, 8, 45}, {0x009, 8, 53}, {0x008, 8, 57}, {0x007, 8, 30},
    {0x006, 8, 46}, {0x005, 8, 54}, {0x004, 8, 58}, {0x007, 9, 31}, {0x006, 9, 47}, {0x005, 9, 55},
    {0x004, 9, 59}, {0x003, 9, 27}, {0x002, 9, 39},
};

// table B-4, motion vector codes
static const VlcCode kMotionCodes[33] = {
    {0x001, 1, 0}, {0x002, 3, 1}, {0x003, 3, -1}, {0x002, 4, 2}, {0x003, 4, -2}, {0x002, 5, 3},
    {0x003, 5, -3}, {0x006, 7, 4}, {0x007, 7, -4}, {0x00A, 8, 5}, {0x00B, 8, -5}, {0x008, 8, 6},
    {0x009, 8, -6}, {0x006, 8, 7}, {0x007, 8, -7}, {0x016, 10, 8}, {0x017, 10, -8}, {0x014, 10, 9},
    {0x015, 10, -9}, {0x012, 10, 10}, {0x013, 10, -10}, {0x022, 11, 11}, {0x023, 11, -11},
    {0x020, 11, 12}, {0x021, 11, -12}, {0x01E, 11, 13}, {0x01F, 11, -13}, {0x01C, 11, 14},
    {0x01D, 11, -14}, {0x01A, 11, 15}, {0x01B, 11, -15}, {0x018, 11, 16}, {0x019, 11, -16},
};

// table B-5c..f, dct_coeff_next without the "1x" codes: "10" is end_of_block and "11s" is
// (run 0, level 1); as the FIRST coefficient of a non-intra block "1s" is (0,1).  The escape
// code is 000001 followed by a 6-bit run and an 8- or 16-bit level.
static const int kDctEscapeCode = 0x01, kDctEscapeLen = 6;
static const DctCode kDctCodes[110] = {
    {0x0003, 3, 1, 1}, {0x0004, 4, 0, 2}, {0x0005, 4, 2, 1}, {0x0005, 5, 0, 3}, {0x0006, 5, 4, 1},
    {0x0007, 5, 3, 1}, {0x0004, 6, 7, 1}, {0x0005, 6, 6, 1}, {0x0006, 6, 1, 2}, {0x0007, 6, 5, 1},
    {0x0004, 7, 2, 2}, {0x0005, 7, 9, 1}, {0x0006, 7, 0, 4}, {0x0007, 7, 8, 1}, {0x0020, 8, 13, 1},
    {0x0021, 8, 0, 6}, {0x0022, 8, 12, 1}, {0x0023, 8, 11, 1}, {0x0024, 8, 3, 2}, {0x0025, 8, 1, 3},
    {0x0026, 8, 0, 5}, {0x0027, 8, 10, 1}, {0x0008, 10, 16, 1}, {0x0009, 10, 5, 2},
    {0x000A, 10, 0, 7}, {0x000B, 10, 2, 3}, {0x000C, 10, 1, 4}, {0x000D, 10, 15, 1},
    {0x000E, 10, 14, 1}, {0x000F, 10, 4, 2}, {0x0010, 12, 0, 11}, {0x0011, 12, 8, 2},
    {0x0012, 12, 4, 3}, {0x0013, 12, 0, 10}, {0x0014, 12, 2, 4}, {0x0015, 12, 7, 2},
    {0x0016, 12, 21, 1}, {0x0017, 12, 20, 1}, {0x0018, 12, 0, 9}, {0x0019, 12, 19, 1},
    {0x001A, 12, 18, 1}, {0x001B, 12, 1, 5}, {0x001C, 12, 3, 3}, {0x001D, 12, 0, 8},
    {0x001E, 12, 6, 2}, {0x001F, 12, 17, 1}, {0x0010, 13, 10, 2}, {0x0011, 13, 9, 2},
    {0x0012, 13, 5, 3}, {0x0013, 13, 3, 4}, {0x0014, 13, 2, 5}, {0x0015, 13, 1, 7},
    {0x0016, 13, 1, 6}, {0x0017, 13, 0, 15}, {0x0018, 13, 0, 14}, {0x0019, 13, 0, 13},
    {0x001A, 13, 0, 12}, {0x001B, 13, 26, 1}, {0x001C, 13, 25, 1}, {0x001D, 13, 24, 1},
    {0x001E, 13, 23, 1}, {0x001F, 13, 22, 1}, {0x0010, 14, 0, 31}, {0x0011, 14, 0, 30},
    {0x0012, 14, 0, 29}, {0x0013, 14, 0, 28}, {0x0014, 14, 0, 27}, {0x0015, 14, 0, 26},
    {0x0016, 14, 0, 25}, {0x0017, 14, 0, 24}, {0x0018, 14, 0, 23}, {0x0019, 14, 0, 22},
    {0x001A, 14, 0, 21}, {0x001B, 14, 0, 20}, {0x001C, 14, 0, 19}, {0x001D, 14, 0, 18},
    {0x001E, 14, 0, 17}, {0x001F, 14, 0, 16}, {0x0010, 15, 0, 40}, {0x0011, 15, 0, 39},
    {0x0012, 15, 0, 38}, {0x0013, 15, 0, 37}, {0x0014, 15, 0, 36}, {0x0015, 15, 0, 35},
    {0x0016, 15, 0, 34}, {0x0017, 15, 0, 33}, {0x0018, 15, 0, 32}, {0x0019, 15, 1, 14},
    {0x001A, 15, 1, 13}, {0x001B, 15, 1, 12}, {0x001C, 15, 1, 11}, {0x001D, 15, 1, 10},
    {0x001E, 15, 1, 9}, {0x001F, 15, 1, 8}, {0x0010, 16, 1, 18}, {0x0011, 16, 1, 17},
    {0x0012, 16, 1, 16}, {0x0013, 16, 1, 15}, {0x0014, 16, 6, 3}, {0x0015, 16, 16, 2},
    {0x0016, 16, 15, 2}, {0x0017, 16, 14, 2}, {0x0018, 16, 13, 2}, {0x0019, 16, 12, 2},
    {0x001A, 16, 11, 2}, {0x001B, 16, 31, 1}, {0x001C, 16, 30, 1}, {0x001D, 16, 29, 1},
    {0x001E, 16, 28, 1}, {0x001F, 16, 27, 1},
};

// scan order of the 64 coefficients (ISO 11172-2 2.4.4.1), raster index for scan position n
static const uint8_t kZigZag[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// default intra quantiser matrix in raster order (ISO 11172-2 2.4.3.2)
static const uint8_t kDefaultIntraQ[64] = {
    8,  16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37, 19, 22, 26, 27, 29, 34,
    34, 38, 22, 22, 26, 27, 29, 34, 37, 40, 22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32,
    35, 40, 48, 58, 26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83};

}  // namespace efx
