// Host-only pieces of the library - the static ray tree (host_tree.h), the sphere grouping
// (hier_build.h) and the MT19937 jump polynomials (mt_jump.h) - in a program of their own, so that
// they can run under AddressSanitizer / UndefinedBehaviorSanitizer on a machine without a GPU
// (tests/test_host_sanitize.py builds it with the sanitizers on the host side only and compares what
// it prints with the Python host layer and with the library's own exports).  No device call is made.
//
//   host_checks tree REFL REFR          -> one line per array: name v0 v1 ...
//   host_checks hier N SEED             -> grouping of N pseudo-random spheres (+ one huge ground sphere)
//   host_checks hierfile FILE N         -> grouping of the N spheres stored as doubles [N][4] in FILE
//   host_checks jump K                  -> 624 words of x^(2^K) mod phi
//   host_checks grid N SEED             -> uniform grid and point-light occluder map of the same spheres: sizes and checksums
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../raytracing_amd/csrc/host_tree.h"
#include "../../raytracing_amd/csrc/mt_jump.h"
#include "../../raytracing_amd/csrc/hier_build.h"
#include "../../raytracing_amd/csrc/grid_build.h"
#include "../../raytracing_amd/csrc/lightmap_build.h"

template <typename T>
static void line(const char* name, const std::vector<T>& v) {
    printf("%s", name);
    for (const T& x : v) printf(" %lld", (long long)x);
    printf("\n");
}

static int print_hierarchy(const std::vector<double>& g) {
    const rt::HostHierarchy H = rt::build_hierarchy(g.data(), g.size() / 4);
    printf("blocks %d\nloose %d\n", H.blocks, H.loose);
    line("ids", H.ids);
    printf("cluster_bound");
    for (int c = 0; c < H.blocks * 8; ++c)
        printf(" %.9g %.9g %.9g %.9g", H.mid[c / 8].cx[c % 8], H.mid[c / 8].cy[c % 8], H.mid[c / 8].cz[c % 8], H.mid[c / 8].re[c % 8]);
    printf("\nblock_bound");
    for (int b = 0; b < H.blocks; ++b)
        printf(" %.9g %.9g %.9g %.9g", H.top[b / 8].cx[b % 8], H.top[b / 8].cy[b % 8], H.top[b / 8].cz[b % 8], H.top[b / 8].re[b % 8]);
    printf("\nloose_spheres");
    for (size_t e = 0; e < H.top_ref.size(); ++e)
        if (H.top_ref[e] != rt::HIER_NONE && (H.top_ref[e] & rt::HIER_LOOSE)) printf(" %u", H.top_ref[e] & 0xffffu);
    printf("\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 4 && !strcmp(argv[1], "tree")) {
        rt::HostTree t;
        const std::string problem = rt::build_host_tree(atoi(argv[2]), atoi(argv[3]), t);
        if (!problem.empty()) { printf("error %s\n", problem.c_str()); return 0; }
        line("a", t.a); line("b", t.b); line("parent", t.parent); line("branch", t.branch);
        line("child", t.child); line("level", t.level); line("level_start", t.level_start);
        return 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "hier")) {
        const int n = atoi(argv[2]);
        uint64_t s = (uint64_t)atoll(argv[3]) * 6364136223846793005ull + 1442695040888963407ull;
        auto u = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; };
        std::vector<double> g;
        if (n > 0) g.insert(g.end(), {4.0, 0.0, -1.0e7, 1.0e7});
        for (int i = 1; i < n; ++i) g.insert(g.end(), {2 + 58 * u(), -25 + 50 * u(), 0.3 + 11.7 * u(), 0.15 + 0.45 * u()});
        return print_hierarchy(g);
    }
    if (argc >= 4 && !strcmp(argv[1], "grid")) {
        const int n = atoi(argv[2]);
        uint64_t s = (uint64_t)atoll(argv[3]) * 6364136223846793005ull + 1442695040888963407ull;
        auto u = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; };
        std::vector<double> g;
        if (n > 0) g.insert(g.end(), {4.0, 0.0, -1.0e7, 1.0e7});
        for (int i = 1; i < n; ++i) g.insert(g.end(), {2 + 58 * u(), -25 + 50 * u(), 0.3 + 11.7 * u(), 0.15 + 0.45 * u()});
        const rt::HostGrid G = rt::build_grid(g.data(), g.size() / 4, 2.0);
        long long sum = 0;
        for (uint16_t v : G.items) sum += v;
        printf("dims %d %d %d\nloose %zu\nitems %zu\nitem_sum %lld\n", G.n[0], G.n[1], G.n[2], G.loose.size(), G.items.size(), sum);
        std::vector<char> in_grid(g.size() / 4, 1);
        for (uint16_t i : G.loose) in_grid[i] = 0;
        const double light[3] = {4.0, 20.0, 30.0};
        const rt::HostLightMap M = rt::build_light_map(g.data(), g.size() / 4, light, in_grid);
        sum = 0;
        for (uint16_t v : M.items) sum += v;
        printf("map_usable %d\nmap_always %zu\nmap_items %zu\nmap_item_sum %lld\n", M.usable ? 1 : 0, M.always.size(), M.items.size(), sum);
        return 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "hierfile")) {
        const size_t n = (size_t)atoll(argv[3]);
        std::vector<double> g(n * 4);
        FILE* f = fopen(argv[2], "rb");
        if (!f || fread(g.data(), sizeof(double), g.size(), f) != g.size()) { fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
        fclose(f);
        return print_hierarchy(g);
    }
    if (argc >= 3 && !strcmp(argv[1], "jump")) {
        std::vector<uint32_t> w(624);
        if (!mtjump::pow2_words(atoi(argv[2]), w.data())) { printf("error no polynomial\n"); return 0; }
        line("poly", w);
        return 0;
    }
    fprintf(stderr, "usage: host_checks tree|hier|hierfile|jump|grid ...\n");
    return 2;
}
