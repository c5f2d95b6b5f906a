/* Force-include used ONLY when compiling the unmodified reference src/video.cpp on a
 * desktop host (oracle/_ref build).  It supplies the three things that file expects from
 * the ESP-IDF environment and nothing else.  Test infrastructure - not product code. */
#include <cstring>
#include <cstdint>
#include <unistd.h>
#define vTaskDelay(x) usleep(1000 * (x))
