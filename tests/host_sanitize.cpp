// Harness for tests/test_host_cpu.py::test_host_code_under_sanitizers: index ingestion, the (multi-threaded) dictionary and block
// builders, container and .fur round trips, the three codec conversions, built with the sanitizers.
#include "host/index_io.hpp"
#include <cstdio>
using namespace fg;
int main(int argc, char** argv) {
    HostIndex a;
    load_dump(argv[1], a);
    verify_dict(a.dict, 1);
    DictStats st = dict_stats(a.dict);
    (void)st;
    std::string t = std::string(argv[2]) + "/x.fgidx";
    save_binary(a, t);
    HostIndex b;
    load_binary(t, b);
    verify_dict(b.dict, 7);
    if (a.dict.table != b.dict.table) { puts("table differs"); return 1; }
    if (a.hybrid.blk_words != b.hybrid.blk_words) { puts("blocks differ"); return 1; }
    std::string f = std::string(argv[2]) + "/x.fur";
    save_fur(a, f);
    HostIndex c;
    load_fur(f, c);
    if (a.hybrid.bits != c.hybrid.bits) { puts("fur differs"); return 1; }
    for (int type = 1; type <= 3; ++type) {
        GenericSets g;
        convert_sets(a.hybrid, type, 64, 8, g);
        build_generic_device(g);
    }
    puts("host ok");
}
