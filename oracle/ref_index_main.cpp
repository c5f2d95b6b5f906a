// oracle/_ref indexer harness  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Compiles the *unmodified* reference indexer/indexer.cpp (textually included from where it lies,
// it has no main of its own) and runs its make_index(path) (indexer.cpp:301-308): reads
// <dir>/video.ts, <dir>/video_fwd.ts and <dir>/video_rwd.ts and writes <dir>/video.idx.
//
// Usage: efx_ref_index <dir>
#include "indexer.cpp"  // -I $(REF)/indexer

int main(int argc, char** argv)
{
    if (argc != 2)
        return 2;
    if (!freopen("/dev/null", "w", stdout)) {}
    make_index(string(argv[1]));
    return 0;
}
