// Test driver (host only): the library's host-side helpers under AddressSanitizer and UBSan with damaged inputs --
// mutated and truncated Radiance .hdr files, mutated OBJ text, triangle sets with NaN, infinities, duplicates and
// degenerate extents for the BVH builder (one thread against several).  Prints "ok" when every case ran.
#include "../../path-tracer_amd/csrc/pt_host.cpp"
#include "../../path-tracer_amd/csrc/pt_bvh_sah.cpp"

#include <cstdio>
#include <random>

static std::vector<uint8_t> good_hdr(uint32_t w, uint32_t h, std::mt19937 &rng)
{
    std::string head = "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y " + std::to_string(h) + " +X " + std::to_string(w) + "\n";
    std::vector<uint8_t> out(head.begin(), head.end());
    for (uint32_t y = 0; y < h; y++) {
        if (y & 1) {  // run-length coded row
            out.insert(out.end(), {2, 2, (uint8_t)(w >> 8), (uint8_t)(w & 255)});
            for (int comp = 0; comp < 4; comp++) {
                uint32_t x = 0;
                while (x < w) {
                    const uint32_t n = std::min<uint32_t>(w - x, 1 + rng() % 100);
                    if (rng() & 1) { out.push_back((uint8_t)(128 + std::min<uint32_t>(n, 127))); out.push_back((uint8_t)rng()); x += std::min<uint32_t>(n, 127); }
                    else { out.push_back((uint8_t)n); for (uint32_t k = 0; k < n; k++) out.push_back((uint8_t)rng()); x += n; }
                }
            }
        } else {      // flat row with an old-style repeat marker
            bool after_marker = true;  // two markers in a row would multiply the second count by 256
            for (uint32_t x = 0; x < w; x++) {
                if (!after_marker && x + 5 <= w && rng() % 7 == 0) { out.insert(out.end(), {1, 1, 1, 4}); x += 3; after_marker = true; }
                else { out.insert(out.end(), {(uint8_t)(3 + rng() % 250), (uint8_t)rng(), (uint8_t)rng(), (uint8_t)(120 + rng() % 16)}); after_marker = false; }
            }
        }
    }
    return out;
}

int main()
{
    std::mt19937 rng(12345);
    // ---- .hdr ----
    int decoded = 0, refused = 0;
    for (int round = 0; round < 40; round++) {
        const uint32_t w = 8 + rng() % 300, h = 1 + rng() % 6;
        const std::vector<uint8_t> base = good_hdr(w, h, rng);
        uint32_t gw = 0, gh = 0;
        std::vector<float> px((size_t)w * h * 4);
        if (pt_hdr_decode(base.data(), base.size(), px.data(), px.size() / 4, &gw, &gh) != PT_OK || gw != w || gh != h) {
            std::printf("good file refused: %s\n", pt_host_error());
            return 1;
        }
        for (int m = 0; m < 250; m++) {
            std::vector<uint8_t> f = base;
            const int kind = rng() % 4;
            if (kind == 0) f.resize(rng() % (f.size() + 1));
            else if (kind == 1) for (int k = 0; k < 1 + (int)(rng() % 4); k++) f[rng() % f.size()] = (uint8_t)rng();
            else if (kind == 2) f.insert(f.begin() + rng() % f.size(), (uint8_t)rng());
            else f.erase(f.begin() + rng() % f.size());
            std::vector<uint8_t> exact(f);  // exact-size heap block: a read past the end is an ASan error
            uint32_t ww = 0, hh = 0;
            int rc = pt_hdr_decode(exact.data(), exact.size(), nullptr, 0, &ww, &hh);
            if (rc == PT_OK && (uint64_t)ww * hh <= (1u << 22)) {
                std::vector<float> out((size_t)ww * hh * 4);
                rc = pt_hdr_decode(exact.data(), exact.size(), out.data(), out.size() / 4, &ww, &hh);
            }
            (rc == PT_OK ? decoded : refused)++;
        }
    }
    std::printf("hdr: %d decoded, %d refused\n", decoded, refused);
    // ---- OBJ ----
    const std::string obj = "# cube corner\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nvn 0 0 1\nvn 1 0 0\nf 1//1 2//1 3//2\nf 1 2 4\nf -1//-1 -2//-2 -3 1/1/1\n";
    char path[] = "/tmp/ptrt_fuzz_XXXXXX";
    const int fd = mkstemp(path);
    if (fd < 0) return 2;
    int loaded = 0, bad = 0;
    for (int m = 0; m < 1500; m++) {
        std::string t = obj;
        for (int k = 0; k < (int)(rng() % 4); k++) {
            const int kind = rng() % 3;
            if (kind == 0) t[rng() % t.size()] = "0123456789-/ .vfn\n#e+"[rng() % 21];
            else if (kind == 1) t.insert(rng() % t.size(), 1, "0123456789-/ .vfn\n"[rng() % 18]);
            else t.erase(rng() % t.size(), 1 + rng() % 3);
        }
        FILE *fp = std::fopen(path, "wb");
        std::fwrite(t.data(), 1, t.size(), fp);
        std::fclose(fp);
        uint32_t n = 0;
        int rc = pt_obj_load(path, nullptr, 0, &n);
        if (rc == PT_OK && n) {
            std::vector<PtTriangle> tris(n);
            rc = pt_obj_load(path, tris.data(), n, &n);
        }
        (rc == PT_OK ? loaded : bad)++;
    }
    std::remove(path);
    std::printf("obj: %d loaded, %d refused\n", loaded, bad);
    // ---- BVH ----
    for (int round = 0; round < 12; round++) {
        const uint32_t n = round < 8 ? 1 + rng() % 400 : 70000 + rng() % 5000;  // the big ones take the threaded path
        std::vector<PtTriangle> tris(n);
        std::uniform_real_distribution<float> u(-3.0f, 3.0f);
        for (PtTriangle &t : tris) {
            std::memset(&t, 0, sizeof t);
            const float c[3] = {u(rng), u(rng), u(rng)};
            for (int k = 0; k < 3; k++) { t.a[k] = c[k] + 0.1f * u(rng); t.b[k] = c[k] + 0.1f * u(rng); t.c[k] = c[k] + 0.1f * u(rng); }
        }
        for (int k = 0; k < 6 && n > 10; k++) {
            PtTriangle &t = tris[rng() % n];
            const int kind = rng() % 5;
            if (kind == 0) t.a[rng() % 3] = NAN;
            else if (kind == 1) t.b[rng() % 3] = INFINITY;
            else if (kind == 2) t.c[rng() % 3] = -INFINITY;
            else if (kind == 3) t = tris[0];
            else { t.a[0] = t.b[0] = t.c[0] = 3.0e38f; }
        }
        if (round == 3) for (PtTriangle &t : tris) t = tris[0];  // all identical: no split possible
        std::vector<PtNode> n1(2 * (size_t)n - 1), n4(2 * (size_t)n - 1);
        std::vector<uint32_t> i1(n), i4(n);
        setenv("PTRT_BUILD_THREADS", "1", 1);
        const int64_t u1 = pt_bvh_build(tris.data(), n, n1.data(), i1.data());
        setenv("PTRT_BUILD_THREADS", "5", 1);
        const int64_t u4 = pt_bvh_build(tris.data(), n, n4.data(), i4.data());
        if (u1 <= 0 || u1 != u4 || std::memcmp(n1.data(), n4.data(), (size_t)u1 * sizeof(PtNode)) != 0 || i1 != i4) {
            std::printf("bvh round %d: thread counts disagree (%lld vs %lld nodes)\n", round, (long long)u1, (long long)u4);
            return 3;
        }
        // the other builder on the same hostile triangles: in bounds, a permutation, children behind their parent
        std::vector<PtNode> ns(2 * (size_t)n - 1);
        std::vector<uint32_t> is(n);
        const int64_t us = pt_bvh_build_sah(tris.data(), n, ns.data(), is.data());
        std::vector<char> seen(n, 0);
        bool good = us >= 1 && us <= 2 * (int64_t)n - 1;
        for (uint32_t k = 0; good && k < n; k++) { good = is[k] < n && !seen[is[k]]; if (good) seen[is[k]] = 1; }
        for (int64_t k = 0; good && k < us; k++) {
            if (ns[k].triangle_count == 0) good = ns[k].left_child_index > k && (int64_t)ns[k].left_child_index + 1 < us;
            else good = (uint64_t)ns[k].left_child_index + ns[k].triangle_count <= n;
        }
        if (!good) {
            std::printf("bvh round %d: the SAH builder's tree is malformed (%lld nodes)\n", round, (long long)us);
            return 4;
        }
        // ... and over padded bounds (hostile pads too: huge, infinite, NaN)
        std::vector<float> pad(n);
        for (float &p : pad) p = 0.01f * (float)(rng() % 100);
        if (n > 10) { pad[rng() % n] = 3.0e38f; pad[rng() % n] = INFINITY; pad[rng() % n] = NAN; }
        const int64_t up = pt_bvh_build_sah_padded(tris.data(), n, pad.data(), ns.data(), is.data());
        std::fill(seen.begin(), seen.end(), 0);
        good = up >= 1 && up <= 2 * (int64_t)n - 1;
        for (uint32_t k = 0; good && k < n; k++) { good = is[k] < n && !seen[is[k]]; if (good) seen[is[k]] = 1; }
        for (int64_t k = 0; good && k < up; k++) {
            if (ns[k].triangle_count == 0) good = ns[k].left_child_index > k && (int64_t)ns[k].left_child_index + 1 < up;
            else good = (uint64_t)ns[k].left_child_index + ns[k].triangle_count <= n;
        }
        if (!good) {
            std::printf("bvh round %d: the padded SAH builder's tree is malformed (%lld nodes)\n", round, (long long)up);
            return 5;
        }
    }
    std::printf("bvh: thread counts agree\nok\n");
    return 0;
}
