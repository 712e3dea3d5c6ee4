#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "host_bvh.h"
using namespace rt;
int main() {
    const int n = 150000;
    std::vector<RtShape> shapes(n);
    std::mt19937_64 g(5);
    std::uniform_real_distribution<double> u(-1, 1), sc(0.002, 0.02);
    for (int i = 0; i < n; ++i) {
        RtShape& s = shapes[i];
        std::memset(&s, 0, sizeof(s));
        s.type = RT_SHAPE_SPHERE;
        double t[3] = {u(g), u(g), u(g)}, k[3] = {sc(g), sc(g), sc(g)};
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { s.transform[r * 4 + c] = (r == c) ? (r < 3 ? k[r] : 1.0) : 0.0; s.inv_transform[r * 4 + c] = (r == c) ? (r < 3 ? 1.0 / k[r] : 1.0) : 0.0; }
        for (int r = 0; r < 3; ++r) { s.transform[r * 4 + 3] = t[r]; s.inv_transform[r * 4 + 3] = -t[r] / k[r]; }
    }
    RtSceneDesc d{};
    d.shapes = shapes.data();
    d.n_shapes = n;
    d.shutter_time = 0.5;
    HostBvh ref = build_bvh(d);
    HostBvh trav = build_traversal_bvh(permuted_boxes(ref.shape_boxes, ref.order), 31);
    Bvh4 b4 = to_node4(trav, 3.0);
    std::printf("ref nodes %zu depth %d, trav nodes %zu depth %d, wide nodes %zu depth %d\n", ref.nodes.size(), ref.max_depth, trav.nodes.size(), trav.max_depth, b4.nodes.size(), b4.max_depth);
    return 0;
}
