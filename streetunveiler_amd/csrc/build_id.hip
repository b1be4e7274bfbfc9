// build_id.hip -- which sources this library was built from.  streetunveiler_amd/build.py passes -DSR_SOURCE_DIGEST="<sha256 over csrc/*, include/*.h
// and the build script, 16 hex digits>"; build() rebuilds when the digest inside the shipped .so differs from the one of the sources next to
// it, and the Python loader refuses a library whose digest is not the sources' (a prebuilt .so travels to the GPU box: this is what ties it
// to the tree it travels with).
#include "../../include/surfel_raster.h"

#ifndef SR_SOURCE_DIGEST
#define SR_SOURCE_DIGEST "unknown"
#endif

extern "C" const char* sr_source_digest(void) {
    static const char tag[] = "SR_SOURCE_DIGEST=" SR_SOURCE_DIGEST;   // (the tag is what build.py looks for in the file without loading it)
    return tag + sizeof("SR_SOURCE_DIGEST=") - 1;
}
