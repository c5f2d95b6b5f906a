// video.h -- DROP-IN for the reference's src/video.h (class Frame, video_init / video_reset / video_pause,
// push_video, push_audio, the overlay buffer; reference src/video.h:36-55).  See player.h in this directory.
#ifndef EFX_DROPIN_VIDEO_H
#define EFX_DROPIN_VIDEO_H
#define EFX_PLAYER_USE_REFERENCE_PLATFORM
#include "../efx_player.hpp"
#endif
