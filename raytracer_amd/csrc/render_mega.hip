// Megakernel variant of the render path: one lane = one pixel, one wave = an 8x8 pixel tile, the whole Whitted tree of
// a pixel evaluated by its lane with a bounded pending-ray stack.  Kept as the simple single-launch schedule
// (RtRenderParams.reserved = 1) and as an independent cross-check of the wavefront schedule of render_wavefront.hip;
// both call the same device functions (device_rt.h).
#include <hip/hip_runtime.h>

#include "device_rt.h"
#include "render_kernel.h"

namespace rt {

namespace {

#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 4   // occupancy target handed to the register allocator (measured: DESIGN.md)
#endif
constexpr int kBlock = 256;            // 4 waves; each wave owns an 8x8 pixel tile, a block a 32x8 strip

// ------------------------------------------------------------------------------------------ shading (shading.h)
// shading.h:42-75, the recursion through transparent occluders unrolled into a loop (each level multiplies the
// running transmission by its own tint; the < 0.001 cut-off is applied to the level's own factor, as there).
template <int F>
__device__ D3 shadow_transmission(const DScene& S, const DParams& P, Ray sr, double dist, Counters& C, TravStack& stack) {
    D3 total = mk(1.0, 1.0, 1.0);
    for (int seg = 0; seg < kShadowMaxSegments; ++seg) {
        Cand c;
        if (!closest_hit<F>(S, P, sr, 0.001, dist - 0.001, c, C, stack)) return total;
        Hit h;
        fill_hit<F>(S, P, sr, c, h);
        const DMaterial& M = S.materials[h.material];
        if (!(M.transparency > 0.0)) return mk(0, 0, 0);
        double n1 = h.front ? 1.0 : M.ior, n2 = h.front ? M.ior : 1.0;
        double eta = n1 / n2;
        double cos_i = -dot(sr.d, h.n);
        double sin2 = eta * eta * (1.0 - cos_i * cos_i);
        if (sin2 > 1.0) return mk(0, 0, 0);
        double r0 = ((n1 - n2) / (n1 + n2));
        r0 = r0 * r0;
        double refl = r0 + (1.0 - r0) * pow((1.0 - cos_i), 5.0);  // shadow_schlick, shading.h:24-28
        D3 local = cmul(mk(1.0, 1.0, 1.0), ld3(M.diffuse)) * (1.0 - refl);
        if (length(local) < 0.001) return mk(0, 0, 0);
        total = cmul(total, local);
        sr.o = h.p + sr.d * 0.001;
        dist = dist - h.t;
    }
    return total;
}

// shading.h:79-101
template <int F>
__device__ D3 light_visibility(const DScene& S, const DParams& P, const DLight& L, D3 Pt, D3 N, double time, Rng& rng, Counters& C,
                               TravStack& stack) {
    if (!P.shadows) return mk(1.0, 1.0, 1.0);
    D3 acc = mk(0, 0, 0);
    for (int i = 0; i < P.shadow_samples; ++i) {
        D3 lp = ld3(L.pos);
        if (L.radius != 0.0) lp = lp + rng.on_unit_sphere() * L.radius;  // shading.h:31-36
        D3 d = lp - Pt;
        double dist = length(d);
        d = normalize(d);
        Ray sr{Pt + N * P.eps_shadow, d, time};
        acc = acc + shadow_transmission<F>(S, P, sr, dist, C, stack);
    }
    return acc * (1.0 / P.shadow_samples);
}

// shading.h:104-175 (ambient + diffuse) and shading.h:181-217 (specular); visibility is evaluated separately for
// each, as the reference does.
template <int F>
__device__ void shade_local(const DScene& S, const DParams& P, const Hit& h, const Ray& view, Rng& rng, Counters& C,
                            TravStack& stack, D3& ad, D3& spec) {
    const DMaterial& M = S.materials[h.material];
    D3 base;
    if (Feat<F>::tex && M.texture >= 0) {
        const DTexture& T = S.textures[M.texture];
        int x = to_int(h.u * (T.width - 1)), y = to_int((1.0 - h.v) * (T.height - 1));
        x = imin(imax(x, 0), T.width - 1);
        y = imin(imax(y, 0), T.height - 1);
        Px p = tex_get(T, x, y);
        base = mk(p.r / 255.0, p.g / 255.0, p.b / 255.0);
    } else {
        base = ld3(M.diffuse);
    }
    D3 colour = cmul(cmul(ld3(M.ambient), ld3(P.ambient_light)), base);
    const D3 Pt = h.p, N = normalize(h.n);
    for (int li = 0; li < S.n_lights; ++li) {
        const DLight& L = S.lights[li];
        D3 vis = light_visibility<F>(S, P, L, Pt, N, view.time, rng, C, stack);
        if (vis.x > 0 || vis.y > 0 || vis.z > 0) {
            D3 Lr = ld3(L.pos) - Pt;
            double falloff = 1.0 / dot(Lr, Lr);
            D3 Ld = normalize(Lr);
            D3 I = ld3(L.intensity) * falloff * P.exposure;
            double ldn = smax(0.0, dot(Ld, N));
            D3 diff = cmul(base, I) * ldn;
            diff = diff * (1.0 - M.transparency);
            colour = colour + cmul(diff, vis);
        }
    }
    ad = colour;
    const D3 V = normalize(view.o - Pt);
    D3 sp = mk(0, 0, 0);
    for (int li = 0; li < S.n_lights; ++li) {
        const DLight& L = S.lights[li];
        D3 vis = light_visibility<F>(S, P, L, Pt, N, view.time, rng, C, stack);
        if (length(vis) > 0) {
            D3 Lr = ld3(L.pos) - Pt;
            double falloff = 1.0 / dot(Lr, Lr);
            D3 Ld = normalize(Lr);
            D3 Hv = normalize(Ld + V);
            D3 I = ld3(L.intensity) * falloff * P.exposure;
            double hdn = smax(0.0, dot(Hv, N));
            D3 part = cmul(ld3(M.specular), I) * exp(M.shininess * log(hdn));  // fast_pow, shading.h:177
            sp = sp + cmul(part, vis);
        }
    }
    spec = sp;
}

struct Pending {   // one not-yet-traced ray of the Whitted tree: colour contribution = weight (x) ray_colour(ray, depth)
    D3 o, d, w;
    int depth;
};


// ray_colour(ray, depth) of tracer.h:147-295 as a loop over a per-lane stack of pending rays.  The colour of a hit is
// linear in the colours of its child rays, so each child carries the product of the weights above it and every
// visited hit adds weight (x) local term to the pixel.
template <int F>
__device__ D3 trace(const DScene& S, const DParams& P, const Ray& primary, Rng& rng, Counters& C, TravStack& tstack) {
    Pending stack[kRayStackDepth];
    int sp = 0;
    D3 accum = mk(0, 0, 0);
    stack[sp++] = Pending{primary.o, primary.d, mk(1.0, 1.0, 1.0), P.max_depth};
    const double eps = P.eps_scene;
    const double time = primary.time;
    while (sp > 0) {
        Pending cur = stack[--sp];
        if (cur.depth <= 0) continue;  // tracer.h:149
        Ray r{cur.o, cur.d, time};
        Cand c;
        if (!closest_hit<F>(S, P, r, eps, 100000.0, c, C, tstack)) {
            D3 bg = (Feat<F>::env && S.env_present) ? env_sample(S, r.d) : ld3(P.background);  // tracer.h:282-293
            accum = accum + cmul(cur.w, bg);
            continue;
        }
        Hit h;
        fill_hit<F>(S, P, r, c, h);
        if (P.normals) {  // tracer.h:157-161
            accum = accum + cmul(cur.w, (h.n + mk(1, 1, 1)) * 0.5);
            continue;
        }
        D3 ad, spec;
        shade_local<F>(S, P, h, r, rng, C, tstack, ad, spec);
        const DMaterial& M = S.materials[h.material];
        const bool transparent = M.transparency > 0;
        bool has_refl = M.reflectivity > 0;
        if (transparent && P.fresnel) has_refl = true;

        double reflect_prob = M.reflectivity, transmit_prob = M.transparency;
        bool do_refract = false, tir_mirror = false;
        D3 refr_dir = mk(0, 0, 0), Vin = mk(0, 0, 0), Nh = mk(0, 0, 0);
        if (transparent) {  // tracer.h:226-266 with compute_refraction tracer.h:100-143
            Vin = normalize(r.d);
            Nh = normalize(h.n);
            double n1 = h.front ? 1.0 : M.ior, n2 = h.front ? M.ior : 1.0;
            double eta = n1 / n2;
            double cos_i = -dot(Vin, Nh);
            double sin2 = eta * eta * (1.0 - cos_i * cos_i);
            if (sin2 > 1.0) {
                transmit_prob = 0.0;
                reflect_prob = 1.0;
                tir_mirror = !has_refl;
            } else {
                double cos_t = sqrt(1.0 - sin2);
                refr_dir = (eta * Vin) + (eta * cos_i - cos_t) * Nh;
                do_refract = true;
                if (P.fresnel) {
                    double r0 = ((n1 - n2) / (n1 + n2)) * ((n1 - n2) / (n1 + n2));  // schlick, tracer.h:88-97
                    double x = 1.0 - cos_i;
                    reflect_prob = r0 + (1.0 - r0) * (x * x * x * x * x);
                    transmit_prob = 1.0 - reflect_prob;
                }
            }
        }
        // tracer.h:268-279
        D3 local;
        double refl_weight;
        if (transparent) { local = spec; refl_weight = reflect_prob; }
        else { local = ad * (1.0 - M.reflectivity) + spec; refl_weight = M.reflectivity; }
        accum = accum + cmul(cur.w, local);

        const int child_depth = cur.depth - 1;
        if (child_depth <= 0) continue;  // children would return black without tracing (tracer.h:149)
        if (has_refl) {  // tracer.h:176-219
            int samples = P.glossy_samples;
            if (cur.depth < P.max_bounces) samples = 1;
            double rough = 1.0 / sqrt(M.shininess);
            D3 V = normalize(r.d);
            D3 perfect = normalize(reflect(V, h.n));
            D3 wr = cur.w * refl_weight;
            if (M.is_metal) wr = cmul(wr, ld3(M.diffuse));
            D3 org = h.p + h.n * eps;
            if (samples > 0) {
                wr = wr * (1.0 / samples);
                for (int i = 0; i < samples; ++i) {
                    D3 off = rng.in_unit_sphere() * rough;
                    D3 target = normalize(perfect + off);
                    if (dot(target, h.n) > 0 && sp < kRayStackDepth) stack[sp++] = Pending{org, target, wr, child_depth};
                }
            } else if (sp < kRayStackDepth) {
                stack[sp++] = Pending{org, perfect, wr, child_depth};
            }
        }
        if (tir_mirror && sp < kRayStackDepth)  // tracer.h:258-264 (not tinted for metals)
            stack[sp++] = Pending{h.p + Nh * eps, normalize(reflect(Vin, Nh)), cur.w * reflect_prob, child_depth};
        if (do_refract && sp < kRayStackDepth)  // tracer.h:239-245: starts at the hit point, tinted by diffuse
            stack[sp++] = Pending{h.p, normalize(refr_dir), cmul(cur.w * transmit_prob, ld3(M.diffuse)), child_depth};
    }
    return accum;
}

// ------------------------------------------------------------------------------------------ the kernel
template <int F>
__global__ void __launch_bounds__(kBlock, RT_WAVES_PER_SIMD) k_render(const DScene S, const DParams P, uint8_t* __restrict__ out,
                                                   unsigned long long* __restrict__ counters) {
    __shared__ unsigned long long tstack_mem[kTravLdsEntries + 1][kBlock];   // + 1: discard row of push_sorted
    TravStack tstack;
    tstack.lds = &tstack_mem[0][threadIdx.x];
    tstack.stride = kBlock;

    // XCD-aware block order: hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own
    // L2); remap so that every XCD works on one contiguous run of image tiles and re-uses the BVH nodes it pulled in.
    const int nblk = gridDim.x;
    const int q = nblk / 8, rem = nblk % 8;
    const int xcd = blockIdx.x % 8, within = blockIdx.x / 8;
    const int blk = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + within;

    const int blocks_x = (P.width + 31) / 32;
    const int bx = blk % blocks_x, by = blk / blocks_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = bx * 32 + wave * 8 + (lane & 7);
    const int lr = by * 8 + (lane >> 3);   // row inside this part's compact slice
    const bool active = (x < P.width) && (lr < P.part_rows);

    Counters C{0u, 0u, 0u, 0u};
    if (active) {
        // compact row -> frame row: tile k of the slice is frame tile k*part_count + part_index
        const int y = ((lr / P.tile_rows) * P.part_count + P.part_index) * P.tile_rows + (lr % P.tile_rows);
        D3 acc = mk(0, 0, 0);
        for (int s = 0; s < P.spp; ++s) {  // main.cpp:288-306
            Rng rng;
            rng.d0 = (P.rng_mode == RT_RNG_D0);
            rng.s = (P.seed * 0x9E3779B97F4A7C15ull) ^
                    (((unsigned long long)y * P.width + x) * 0xD1B54A32D192ED03ull + (unsigned long long)s * 0x8CB92BA72F3D8DD7ull);
            float ru = (float)rng.uniform(), rv = (float)rng.uniform();
            float px = ((float)x + ru) / (float)P.width;
            float py = ((float)y + rv) / (float)P.height;
            double time = rng.uniform() * S.shutter_time;
            Ray ray = generate_ray(S.cam, px, py, time, rng);
            acc = acc + trace<F>(S, P, ray, rng, C, tstack);
        }
        D3 avg = tonemap(acc * (1.0 / P.spp), P.tonemap);  // main.cpp:308-316
        uint8_t* o = out + ((size_t)lr * P.width + x) * 3;
        o[0] = quantise(avg.x);
        o[1] = quantise(avg.y);
        o[2] = quantise(avg.z);
    }
    if (counters) {  // one atomic per wave and counter
        unsigned int rays = C.rays, nodes = C.nodes, prims = C.prims, visits = C.visits;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            rays += __shfl_down(rays, off);
            nodes += __shfl_down(nodes, off);
            prims += __shfl_down(prims, off);
            visits += __shfl_down(visits, off);
        }
        if (lane == 0) {
            atomicAdd(&counters[0], (unsigned long long)rays);
            if (P.count_tests) {
                atomicAdd(&counters[1], (unsigned long long)nodes);
                atomicAdd(&counters[2], (unsigned long long)prims);
                atomicAdd(&counters[4], (unsigned long long)visits);
            }
        }
    }
}

}  // namespace

hipError_t launch_render_mega(const DScene& scene, const DParams& params, uint8_t* d_rgb8, unsigned long long* d_counters,
                         hipStream_t stream) {
    const int blocks_x = (params.width + 31) / 32;
    const int blocks_y = (params.part_rows + 7) / 8;
    const int nblk = blocks_x * blocks_y;
    if (nblk <= 0) return hipSuccess;
    // smallest compiled variant that covers the scene's features
    constexpr int kSpheres = RT_FEAT_SPHERE;
    constexpr int kAnalytic = RT_FEAT_SPHERE | RT_FEAT_QUAD | RT_FEAT_CUBE;
    constexpr int kFull = kAnalytic | RT_FEAT_COMPLEX | RT_FEAT_TEXTURE | RT_FEAT_ENV;
    const int f = scene.features;
    if ((f & ~kSpheres) == 0)
        hipLaunchKernelGGL(k_render<kSpheres>, dim3(nblk), dim3(kBlock), 0, stream, scene, params, d_rgb8, d_counters);
    else if ((f & ~kAnalytic) == 0)
        hipLaunchKernelGGL(k_render<kAnalytic>, dim3(nblk), dim3(kBlock), 0, stream, scene, params, d_rgb8, d_counters);
    else
        hipLaunchKernelGGL(k_render<kFull>, dim3(nblk), dim3(kBlock), 0, stream, scene, params, d_rgb8, d_counters);
    return hipGetLastError();
}

}  // namespace rt
