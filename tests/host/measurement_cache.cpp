// Host unit test (no device): VisionMeasurement's flat cache (eqvio_amd/host/VIOFilter.hpp) must follow edits of the public
// camCoordinates map — same size and same end ids, different pixels or a different interior id — and copies must not inherit a stale cache.
#include "VIOFilter.hpp"
#include <cstdio>
#include <cstdlib>

using namespace eqvio_amd;

#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::fprintf(stderr, "FAILED %s (line %d)\n", #cond, __LINE__); \
            std::exit(1);                                                   \
        }                                                                   \
    } while (0)

int main() {
    VisionMeasurement m;
    for (int id : {3, 7, 11, 19})
        m.camCoordinates[id] = {10.0 * id, 20.0 * id};
    CHECK((m.flatIds() == std::vector<int>{3, 7, 11, 19}));
    CHECK(m.flatY()[2] == 70.0 && m.flatY()[3] == 140.0);
    // new pixel values under the same ids
    m.camCoordinates[7] = {1.5, 2.5};
    CHECK(m.flatY()[2] == 1.5 && m.flatY()[3] == 2.5);
    // an interior id replaced: size, first and last id unchanged
    m.camCoordinates.erase(11);
    m.camCoordinates[13] = {5.0, 6.0};
    CHECK((m.flatIds() == std::vector<int>{3, 7, 13, 19}));
    CHECK(m.flatY()[4] == 5.0 && m.flatY()[5] == 6.0);
    // a copy carries the cache along; editing the copy must not show the original's pixels
    VisionMeasurement c = m;
    c.camCoordinates[19] = {-1.0, -2.0};
    auto v = c.flat();
    CHECK((*v.second)[6] == -1.0 && (*v.second)[7] == -2.0);
    CHECK(m.flatY()[6] == 190.0);
    // a copy made while a Validated guard trusts the original (processVisionData's matchedMeasurement) must not inherit the trust - nothing would ever reset it,
    // and an edit of the copy that keeps its size (a pixel replaced in place) would be served from the stale cache
    {
        VisionMeasurement::Validated guard(m);
        VisionMeasurement d = m; // copied under the guard
        d.camCoordinates[13] = {77.0, 88.0};
        CHECK(d.flatY()[4] == 77.0 && d.flatY()[5] == 88.0);
        VisionMeasurement e2;
        e2 = m; // assigned under the guard
        e2.camCoordinates[3] = {-5.0, -6.0};
        CHECK(e2.flatY()[0] == -5.0 && e2.flatY()[1] == -6.0);
        CHECK(m.flatY()[4] == 5.0); // the guarded original is served from its (validated) cache
    }
    // shrink and grow
    c.camCoordinates.clear();
    CHECK(c.flatIds().empty() && c.flatY().empty());
    c.camCoordinates[1] = {0.25, 0.5};
    CHECK(c.getIds() == std::vector<int>{1} && c.flatY()[1] == 0.5);
    std::puts("ok");
    return 0;
}
