// player.h -- DROP-IN for the reference's src/player.h (class MpegDecoder, reference src/player.h:34-165).
//
// Put this directory in front of the reference's src/ on the include path and leave src/player.cpp and
// src/video.cpp out of the build: the reference's host player (espflix.cpp) and platform layer (streamer.cpp,
// streamer.h: Q, Buffer, the event word, Streamer, threads, printf_nano) compile UNMODIFIED against this file
// and get their pictures from libefx.  See INTEGRATION.md section 1 and tests/dropin_main.cpp.
#ifndef EFX_DROPIN_PLAYER_H
#define EFX_DROPIN_PLAYER_H
#define EFX_PLAYER_USE_REFERENCE_PLATFORM
#include "../efx_player.hpp"
#endif
