// rt_host_scene.hip.h -- host side, part 3 of 4: what rt_scene_upload* does with the reference's arrays -- traversal-order nodes, visit-order triangle records, the
// breadth-first layouts of the work-stack kernel, the fixed-point nodes derived on the device, several meshes as one forest.
#pragma once

namespace {

// Host-side Vector arithmetic for the triangle precompute (cpu:227-229).  This TU is
// compiled with -ffp-contract=off, so these are the same single roundings as on the device.
struct h3 { float x, y, z; };
inline h3 hsub(h3 a, h3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline h3 hcross(h3 a, h3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// Converts the reference's bvhTreeToArray layout (optimized.cu:512-534) into traversal order.
// The reference pops the right child first (cpu:291-292 push left then right), so the
// pre-order here descends right before left.
int build_threaded(rt_ctx *ctx, const rt_mesh *m, std::vector<float4> &lo, std::vector<float4> &hi, std::vector<int> &perm,
                   std::vector<int> &left_of) {
    const int n = m->n_nodes;
    perm.clear();
    left_of.assign(n, -1);                                        // internal nodes: traversal-order index of the LEFT child
    lo.assign(n, make_float4(0, 0, 0, 0));
    hi.assign(n, make_float4(0, 0, 0, 0));
    if (n == 0) return RT_OK;
    struct Item { int ref; int out; int stage; };
    std::vector<char> seen(n, 0);
    std::vector<Item> st;
    int emitted = 0;
    auto node = [&](int i) { return m->bvh_arr10 + (size_t)i * 10; };
    st.push_back({0, -1, 0});
    while (!st.empty()) {
        Item &it = st.back();
        const float *a = node(it.ref);
        if (it.stage == 0) {
            if (seen[it.ref]) return fail(ctx, RT_ERR_INVALID, "bvh_arr10: node %d reached twice (not a tree)", it.ref);
            seen[it.ref] = 1;
            it.out = emitted++;
            const int left = (int)a[0], right = (int)a[1];
            const int ts = (int)a[8], te = (int)a[9];
            if (ts < 0 || te < ts || te > m->n_triangles)
                return fail(ctx, RT_ERR_INVALID, "bvh_arr10: node %d has triangle range [%d,%d) outside [0,%d)", it.ref, ts, te, m->n_triangles);
            lo[it.out] = make_float4(a[2], a[3], a[4], 0);
            hi[it.out] = make_float4(a[5], a[6], a[7], 0);
            if (left == -1 || right == -1) {   // leaf (cpu:287 tests `left` only; the builder sets both or none)
                if (left != -1 || right != -1)
                    return fail(ctx, RT_ERR_INVALID, "bvh_arr10: node %d has exactly one child", it.ref);
                // triangles are re-stored in VISIT order (leaves as the traversal reaches them, ascending inside a
                // leaf, cpu:295), so a triangle's index is its rank in the reference's scan: the strict '<' of
                // cpu:301 keeps, among equal t, the smallest index -- which is what lets sub-ranges of one ray
                // be traversed independently and merged by min over (t, index)
                const int first = (int)perm.size();
                for (int q = ts; q < te; ++q) perm.push_back(q);
                lo[it.out].w = __builtin_bit_cast(float, first);
                hi[it.out].w = __builtin_bit_cast(float, (int)perm.size());
                st.pop_back();
                continue;
            }
            if (left < 0 || left >= n || right < 0 || right >= n)
                return fail(ctx, RT_ERR_INVALID, "bvh_arr10: node %d has a child index out of range", it.ref);
            it.stage = 1;
            st.push_back({right, -1, 0});
        } else if (it.stage == 1) {
            it.stage = 2;
            const int left = (int)a[0];
            left_of[it.out] = emitted;                               // the left subtree starts right behind the right one
            st.push_back({left, -1, 0});
        } else {
            lo[it.out].w = __builtin_bit_cast(float, emitted);   // next node on a box miss: past the subtree
            hi[it.out].w = __builtin_bit_cast(float, -1);
            st.pop_back();
        }
    }
    if (emitted != n) return fail(ctx, RT_ERR_INVALID, "bvh_arr10: %d of %d nodes reachable from the root", emitted, n);
    return RT_OK;
}

// The part of rt_scene_upload after validation of the sphere / light / camera arguments: layout conversion of the mesh (the
// reference's arrays -> traversal-order nodes, visit-order triangle records, breadth-first sibling pairs, refit levels) and
// the uploads.  `sc` carries the spheres, light and camera; rt_mesh_rebuild re-enters here with the tree it built on the device.
constexpr int kQ16AutoNodes = 16384;                                 // RT_TRAVQ_Q16 = -1: from this many nodes on (the node array no longer sits in the L1s)
// (Re)derive the 16-bit fixed-point sibling pairs and the triangle -> leaf table from the breadth-first arrays on the device (rt_qnodes.hip.h), on stream q
// (the upload passes the null stream, as its copies do: creating the context's own stream here would change which hardware queues the sub-frame streams
// get later, profiles/round3/ab_hw_queues_parts.log), joined before returning.  ctx->scene must be final (root box, node arrays); trees the format does not fit keep scene.nodesh = nullptr.
int requantize(rt_ctx *ctx, hipStream_t q) {
    rtk::Scene &sc = ctx->scene;
    sc.nodesh = nullptr; sc.tri2leaf = nullptr; sc.nodesw = nullptr; sc.leaflh = nullptr;
    // (wherever the format fits: with flagged leaves the 4-wide step beats the fixed-point pairs on every tree measured -- 2 019 nodes -8 %, 32 889 -9 %, 358 503 -12 %: profiles/round5/ab_wide_nodes.txt)
    const bool want_qw = ctx->knobs.qw != 0 && ctx->qw_topo_ok && ctx->q16_leaf_shift == 24 && sc.n_nodes + 2 < (1 << 21);   // the quad's payload word: leaves of <= 127 triangles, child << 10 positive; no empty leaf
    if (!(ctx->knobs.q16 == 1 || (ctx->knobs.q16 < 0 && sc.n_nodes >= kQ16AutoNodes) || want_qw) || !ctx->q16_topo_ok || ctx->q16_leaf_shift == 0 || !ctx->travq_ok || !sc.fast_box || sc.mesh_slot < 0 || sc.n_nodes < 3 || sc.n_tris <= 0) return RT_OK;
    int rc;
    if ((rc = ensure(ctx, ctx->nodesh, ((size_t)sc.n_nodes + 2) * 16)) != RT_OK || (rc = ensure(ctx, ctx->tri2leaf, (size_t)sc.n_tris * sizeof(int))) != RT_OK ||
        (rc = ensure(ctx, ctx->leaflh, (size_t)sc.n_tris * 32)) != RT_OK) return rc;
    RT_HIP(ctx, hipMemsetAsync(ctx->tri2leaf.p, 0, (size_t)sc.n_tris * sizeof(int), q));
    const rtk::QGrid g = rtk::q16_grid(sc.root_lo, sc.root_hi);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipMemsetAsync(ctx->nodesh.p, 0, 32, q));            // nodes 0 (padding) and 1 (the root: tested when a ray is emitted)
    if (want_qw && (rc = ensure(ctx, ctx->nodesw, ((size_t)sc.n_nodes + 4) * 32)) != RT_OK) return rc;
    hipLaunchKernelGGL(rtk::qnodes_kernel, dim3((unsigned)((sc.n_nodes + 255) / 256)), dim3(256), 0, q, sc.nodesq, sc.nodesb, sc.n_nodes, g,
                       static_cast<uint4 *>(ctx->nodesh.p), static_cast<int *>(ctx->tri2leaf.p), sc.n_tris, rtk::kQLeafShift, ctx->q16_leaf_shift);
    hipLaunchKernelGGL(rtk::leaflh_kernel, dim3((unsigned)((sc.n_tris + 255) / 256)), dim3(256), 0, q, sc.nodesq, static_cast<const int *>(ctx->tri2leaf.p), sc.n_tris, sc.n_nodes,
                       static_cast<float4 *>(ctx->leaflh.p));
    if (want_qw) {
        const bool dp = ctx->knobs.quad_sel != 0;
        if (dp) {
            const size_t nn = (size_t)sc.n_nodes + 2;
            if ((rc = ensure(ctx, ctx->qdp_parent, nn * 4)) != RT_OK || (rc = ensure(ctx, ctx->qdp_cnt, nn * 4)) != RT_OK || (rc = ensure(ctx, ctx->qdp_g, nn * 16)) != RT_OK ||
                (rc = ensure(ctx, ctx->qdp_ch, nn * 4)) != RT_OK) return rc;
            rtk::QdpArgs a{};
            a.nodesh = static_cast<const uint4 *>(ctx->nodesh.p); a.n_bfs = sc.n_nodes; a.node_shift = rtk::kQNodeShift;
            a.sx = g.sx; a.sy = g.sy; a.sz = g.sz;
            a.parent = static_cast<int *>(ctx->qdp_parent.p); a.cnt = static_cast<int *>(ctx->qdp_cnt.p);
            a.g = static_cast<float4 *>(ctx->qdp_g.p); a.ch = static_cast<uchar4 *>(ctx->qdp_ch.p);
            RT_HIP(ctx, hipMemsetAsync(ctx->qdp_ch.p, 0, nn * 4, q));
            const dim3 grid((unsigned)((sc.n_nodes + 255) / 256));
            hipLaunchKernelGGL(rtk::qdp_init_kernel, grid, dim3(256), 0, q, a);
            hipLaunchKernelGGL(rtk::qdp_up_kernel, grid, dim3(256), 0, q, a);
        }
        hipLaunchKernelGGL(rtk::qquads_kernel, dim3((unsigned)((sc.n_nodes / 2 + 1 + 255) / 256)), dim3(256), 0, q, static_cast<const uint4 *>(ctx->nodesh.p), sc.n_nodes,
                           rtk::kQNodeShift, ctx->q16_leaf_shift, dp ? static_cast<const int *>(ctx->qdp_parent.p) : nullptr, dp ? static_cast<const uchar4 *>(ctx->qdp_ch.p) : nullptr,
                           static_cast<uint4 *>(ctx->nodesw.p));
    }
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipStreamSynchronize(q));
    // the DP's scratch (28 bytes per node) is needed while this function runs only: big trees give it back (a cat-sized one keeps it for the next refit)
    if ((size_t)sc.n_nodes * 28 > (16u << 20)) { ctx->qdp_parent.release(); ctx->qdp_cnt.release(); ctx->qdp_g.release(); ctx->qdp_ch.release(); }
    sc.nodesh = static_cast<const uint4 *>(ctx->nodesh.p); sc.tri2leaf = static_cast<const int *>(ctx->tri2leaf.p); sc.leaflh = static_cast<const float4 *>(ctx->leaflh.p);
    if (want_qw) sc.nodesw = static_cast<const uint4 *>(ctx->nodesw.p);
    sc.qgx = g.gx; sc.qgy = g.gy; sc.qgz = g.gz; sc.qsx = g.sx; sc.qsy = g.sy; sc.qsz = g.sz; sc.qleaf_shift = ctx->q16_leaf_shift;
    return RT_OK;
}

// the triangle ranges of the scene's mesh table when at most ONE mesh has triangles (object position real_obj): a mesh without triangles is an empty range at its place in the order
void mesh_table_single(rtk::Scene &sc, int real_obj) {
    for (int k = 0; k < sc.n_meshes; ++k) sc.mesh[k].tri_begin = sc.mesh[k].obj <= real_obj ? 0 : sc.n_tris;
}

// sc: spheres (with their object ids), light, camera and the mesh table (object ids, materials; sc.mesh_slot = the first mesh object's position or -1) filled in by the caller.
// mesh: the geometry to traverse -- one TriangleMesh as uploaded, or the forest build_forest made of several (tri_offsets[k] = first triangle of table entry k in mesh->indices,
// n_meshes + 1 entries) -- or nullptr.
int install_scene(rt_ctx *ctx, rtk::Scene sc, const rt_mesh *mesh, const std::vector<int> *tri_offsets = nullptr) {
    PhaseClock pc;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->stream_) RT_HIP(ctx, hipStreamSynchronize(ctx->stream_));   // (renders issued on a caller's stream are the caller's to order)
    ctx->have_scene = false;
    ctx->host_mesh_stale = false;                                     // what follows rewrites tri_perm / up_indices
    ctx->tri_perm.clear();
    std::vector<float4> lo, hi, tri, verts;
    std::vector<int4> tidx;
    std::vector<int> left_of;
    if (mesh) {
        if (mesh->object_slot < 0 || mesh->object_slot >= RT_MAX_OBJECTS)   // (validated against the scene's objects by rt_scene_upload_meshes)
            return fail(ctx, RT_ERR_INVALID, "mesh object_slot %d outside [0,%d)", mesh->object_slot, RT_MAX_OBJECTS);
        if (mesh->n_vertices < 0 || mesh->n_triangles < 0 || mesh->n_nodes < 0 || mesh->index_stride < 3)
            return fail(ctx, RT_ERR_INVALID, "bad mesh sizes");
        if ((mesh->n_vertices && !mesh->vertices) || (mesh->n_triangles && !mesh->indices) || (mesh->n_nodes && !mesh->bvh_arr10))
            return fail(ctx, RT_ERR_INVALID, "mesh array pointer is NULL");
        if (mesh->n_nodes >= (1 << 24)) return fail(ctx, RT_ERR_INVALID, "node indices are stored as floats: < 2^24 nodes");
        std::vector<int> perm;
        int rc = build_threaded(ctx, mesh, lo, hi, perm, left_of);
        if (rc != RT_OK) return rc;
        if (perm.size() >= ((size_t)1 << 31)) return fail(ctx, RT_ERR_INVALID, "too many leaf triangles");
        for (int t = 0; t < mesh->n_triangles; ++t) {
            const int32_t *ix = mesh->indices + (size_t)t * mesh->index_stride;
            for (int k = 0; k < 3; ++k)
                if (ix[k] < 0 || ix[k] >= mesh->n_vertices)
                    return fail(ctx, RT_ERR_INVALID, "triangle %d references vertex %d outside [0,%d)", t, ix[k], mesh->n_vertices);
        }
        const int n_int = (int)perm.size();
        ctx->tri_perm = perm;
        ctx->up_indices.resize((size_t)mesh->n_triangles * 3);              // the mesh as uploaded (BVH order): rt_mesh_rebuild starts from it
        std::vector<int4> tup(mesh->n_triangles);
        for (int t = 0; t < mesh->n_triangles; ++t) {
            const int32_t *ix = mesh->indices + (size_t)t * mesh->index_stride;
            for (int k = 0; k < 3; ++k) ctx->up_indices[3 * (size_t)t + k] = ix[k];
            tup[t] = make_int4(ix[0], ix[1], ix[2], 0);
        }
        ctx->n_up_tris = mesh->n_triangles;
        if (int rcu = upload(ctx, ctx->tidx_up, tup.data(), tup.size() * sizeof(int4)); rcu != RT_OK) return rcu;
        tri.resize((size_t)n_int * 3);
        tidx.resize(n_int);
        for (int t = 0; t < n_int; ++t) {
            const int32_t *ix = mesh->indices + (size_t)perm[t] * mesh->index_stride;
            for (int k = 0; k < 3; ++k)
                if (ix[k] < 0 || ix[k] >= mesh->n_vertices)
                    return fail(ctx, RT_ERR_INVALID, "triangle %d references vertex %d outside [0,%d)", t, ix[k], mesh->n_vertices);
            auto V = [&](int i) { return h3{mesh->vertices[3 * (size_t)i], mesh->vertices[3 * (size_t)i + 1], mesh->vertices[3 * (size_t)i + 2]}; };
            const h3 A = V(ix[0]), B = V(ix[1]), C = V(ix[2]);
            const h3 e1 = hsub(B, A), e2 = hsub(C, A), N = hcross(e1, e2);   // cpu:227-229
            tri[3 * (size_t)t + 0] = make_float4(A.x, A.y, A.z, e1.x);
            tri[3 * (size_t)t + 1] = make_float4(e1.y, e1.z, e2.x, e2.y);
            tri[3 * (size_t)t + 2] = make_float4(e2.z, N.x, N.y, N.z);
            tidx[t] = make_int4(ix[0], ix[1], ix[2], 0);
        }
        verts.resize(mesh->n_vertices);
        for (int i = 0; i < mesh->n_vertices; ++i)
            verts[i] = make_float4(mesh->vertices[3 * (size_t)i], mesh->vertices[3 * (size_t)i + 1], mesh->vertices[3 * (size_t)i + 2], 0);
        sc.n_nodes = n_int > 0 ? mesh->n_nodes : 0;
        sc.n_tris = n_int;
        if (sc.n_nodes > 0) { sc.root_lo = lo[0]; sc.root_hi = hi[0]; }
        sc.n_verts = mesh->n_vertices;
        if (tri_offsets) {
            // the forest is laid out so that the traversal reaches the meshes in object order (build_forest): the visit-order triangle array is mesh after mesh
            int cur = 0;
            for (int k = 0; k < sc.n_meshes; ++k) sc.mesh[k].tri_begin = -1;
            for (int t = 0; t < n_int; ++t) {
                while (cur + 1 < sc.n_meshes && perm[t] >= (*tri_offsets)[cur + 1]) ++cur;
                if (perm[t] < (*tri_offsets)[cur]) return fail(ctx, RT_ERR_INTERNAL, "forest layout: triangle %d of an earlier mesh is visited after a later mesh's", perm[t]);
                if (sc.mesh[cur].tri_begin < 0) sc.mesh[cur].tri_begin = t;
            }
            int next = n_int;
            for (int k = sc.n_meshes - 1; k >= 0; --k) { if (sc.mesh[k].tri_begin < 0) sc.mesh[k].tri_begin = next; next = sc.mesh[k].tri_begin; }
        } else {
            mesh_table_single(sc, mesh->object_slot);
        }
    } else {
        mesh_table_single(sc, -1);
    }
    int rc;
    pc.lap("  scene: traversal order, triangle records (host) + the FIRST hipMalloc / copy of the process (runtime: stream = hardware queue, staging)");
    if ((rc = upload(ctx, ctx->node_lo, lo.data(), lo.size() * sizeof(float4))) != RT_OK) return rc;
    pc.lap("  scene: one more hipMalloc + copy");
    if ((rc = upload(ctx, ctx->node_hi, hi.data(), hi.size() * sizeof(float4))) != RT_OK) return rc;
    std::vector<float4> inter(lo.size() * 2);
    for (size_t k = 0; k < lo.size(); ++k) { inter[2 * k] = lo[k]; inter[2 * k + 1] = hi[k]; }
    if ((rc = upload(ctx, ctx->nodes2, inter.data(), inter.size() * sizeof(float4))) != RT_OK) return rc;
    {   // work-stack layout: breadth-first order (the top of the tree is a prefix: LDS staging), children adjacent
        const size_t n = lo.size();
        std::vector<int> order;                                      // order[k] = traversal-order index of breadth-first node k
        std::vector<int> bfs_of(n, -1);
        order.reserve(n);
        if (n) { order.push_back(0); bfs_of[0] = 0; }
        for (size_t k = 0; k < order.size(); ++k) {
            const int x = order[k];
            if (left_of[x] >= 0) {                                    // internal: right child x + 1, left child left_of[x]
                bfs_of[x + 1] = (int)order.size(); order.push_back(x + 1);
                bfs_of[left_of[x]] = (int)order.size(); order.push_back(left_of[x]);
            }
        }
        // index 0 is padding, the root is node 1, so that every sibling pair (2m, 2m + 1) is one aligned 64-byte line
        std::vector<float4> q(2 * (order.size() + 1), make_float4(0, 0, 0, 0));
        std::vector<int> q2t(order.size() + 1, 0);
        for (size_t k = 0; k < order.size(); ++k) {
            const int x = order[k];
            q[2 * (k + 1)] = lo[x]; q[2 * (k + 1) + 1] = hi[x];
            if (left_of[x] >= 0) q[2 * (k + 1)].w = __builtin_bit_cast(float, bfs_of[x + 1] + 1);
            q2t[k + 1] = x;
        }
        if ((rc = upload(ctx, ctx->nodesq, q.data(), q.size() * sizeof(float4))) != RT_OK) return rc;
        // wf_travq's form of the same array (rt_travq.hip.h): box as centre / half extent, payload and kind pre-shifted the way stack
        // and leaf-queue entries carry them; and the scene-wide quantities its box filter needs
        std::vector<float4> qb(q.size(), make_float4(0, 0, 0, 0));
        float bm[3] = {0.f, 0.f, 0.f};
        bool fast = true, travq_ok = true;
        for (size_t k = 0; k < order.size(); ++k) {
            const int x = order[k];
            const float4 l = lo[x], h = hi[x];
            float4 cb = make_float4(rtk::box_centre(l.x, h.x), rtk::box_centre(l.y, h.y), rtk::box_centre(l.z, h.z), 0.f);
            float4 hb = make_float4(rtk::box_half(l.x, h.x), rtk::box_half(l.y, h.y), rtk::box_half(l.z, h.z), 0.f);
            const float v[6] = {l.x, l.y, l.z, h.x, h.y, h.z};
            for (int a = 0; a < 3; ++a) {
                if (!(v[a] <= v[a + 3]) || !(std::fabs(v[a]) < 1e8f) || !(std::fabs(v[a + 3]) < 1e8f)) fast = false;   // also false for NaN
                bm[a] = std::max(bm[a], std::max(std::fabs(v[a]), std::fabs(v[a + 3])));
            }
            if (left_of[x] >= 0) {
                cb.w = __builtin_bit_cast(float, (uint32_t)(bfs_of[x + 1] + 1) << rtk::kQNodeShift);
                hb.w = __builtin_bit_cast(float, (int)0x80000000);
            } else {
                const int first = __builtin_bit_cast(int, l.w), cnt = __builtin_bit_cast(int, h.w) - first;
                if (cnt >= rtk::kQMaxLeaf) travq_ok = false;
                cb.w = __builtin_bit_cast(float, first);
                hb.w = __builtin_bit_cast(float, cnt > 0 && cnt < rtk::kQMaxLeaf ? cnt << rtk::kQLeafShift : 0);
            }
            qb[2 * (k + 1)] = cb; qb[2 * (k + 1) + 1] = hb;
        }
        if ((rc = upload(ctx, ctx->nodesb, qb.data(), qb.size() * sizeof(float4))) != RT_OK) return rc;
        {   // may this tree use the 16-bit fixed-point pairs (rt_qnodes.hip.h)?  Leaf sizes and counts fit the payload word, and every box nests inside its parent's
            bool topo = order.size() >= 3, empty_leaf = false;
            int max_leaf = 0;
            for (size_t x = 0; topo && x < n; ++x) {
                if (left_of[x] < 0) {
                    const int cnt = __builtin_bit_cast(int, hi[x].w) - __builtin_bit_cast(int, lo[x].w);
                    max_leaf = std::max(max_leaf, cnt);
                    if (cnt <= 0) empty_leaf = true;                  // the quads' places 0 and 2 must hold a node; the fixed-point and the float pairs cope with an empty leaf
                    continue;
                }
                for (const int c : {(int)x + 1, left_of[x]}) {
                    const float4 cl = lo[c], ch = hi[c], pl = lo[x], ph = hi[x];
                    if (!(cl.x >= pl.x && cl.y >= pl.y && cl.z >= pl.z && ch.x <= ph.x && ch.y <= ph.y && ch.z <= ph.z)) topo = false;   // also false for NaN
                }
            }
            ctx->q16_topo_ok = topo;
            ctx->qw_topo_ok = topo && !empty_leaf;
            ctx->q16_leaf_shift = rtk::q16_leaf_shift(max_leaf, (long long)(tri.size() / 3));
        }
        sc.bmx = bm[0]; sc.bmy = bm[1]; sc.bmz = bm[2];
        sc.fast_box = fast ? 1 : 0;
        ctx->travq_ok = travq_ok && (uint64_t)tri.size() * 16 < ((uint64_t)1 << 32);   // 32-bit byte offsets into the triangle records
        if ((rc = upload(ctx, ctx->q2thr, q2t.data(), q2t.size() * sizeof(int))) != RT_OK) return rc;
        // levels of the tree (pre-order indices sorted by depth) for the device-side refit (rt_mesh_transform)
        std::vector<int> depth(n, 0), lvl_off, lvl_nodes(n);
        int maxd = 0;
        for (size_t x = 0; x < n; ++x)
            if (left_of[x] >= 0) { depth[x + 1] = depth[left_of[x]] = depth[x] + 1; maxd = std::max(maxd, depth[x] + 1); }
        lvl_off.assign(maxd + 2, 0);
        for (size_t x = 0; x < n; ++x) lvl_off[depth[x] + 1]++;
        for (int d = 0; d <= maxd; ++d) lvl_off[d + 1] += lvl_off[d];
        std::vector<int> fill(lvl_off.begin(), lvl_off.end() - 1);
        for (size_t x = 0; x < n; ++x) lvl_nodes[fill[depth[x]]++] = (int)x;
        ctx->n_levels = n ? maxd + 1 : 0;
        if ((rc = upload(ctx, ctx->left_dev, left_of.data(), left_of.size() * sizeof(int))) != RT_OK) return rc;
        if ((rc = upload(ctx, ctx->lvl_nodes, lvl_nodes.data(), lvl_nodes.size() * sizeof(int))) != RT_OK) return rc;
        if ((rc = upload(ctx, ctx->lvl_off, lvl_off.data(), lvl_off.size() * sizeof(int))) != RT_OK) return rc;
    }
    if ((rc = upload(ctx, ctx->tri, tri.data(), tri.size() * sizeof(float4))) != RT_OK) return rc;
    if ((rc = upload(ctx, ctx->verts, verts.data(), verts.size() * sizeof(float4))) != RT_OK) return rc;
    if ((rc = upload(ctx, ctx->tidx, tidx.data(), tidx.size() * sizeof(int4))) != RT_OK) return rc;
    sc.node_lo = static_cast<const float4 *>(ctx->node_lo.p);
    sc.node_hi = static_cast<const float4 *>(ctx->node_hi.p);
    sc.nodes = static_cast<const float4 *>(ctx->nodes2.p);
    sc.nodesq = static_cast<const float4 *>(ctx->nodesq.p);
    sc.nodesb = static_cast<const float4 *>(ctx->nodesb.p);
    sc.q2thr = static_cast<const int *>(ctx->q2thr.p);
    sc.tri = static_cast<const float4 *>(ctx->tri.p);
    sc.verts = static_cast<const float4 *>(ctx->verts.p);
    sc.tidx = static_cast<const int4 *>(ctx->tidx.p);
    ctx->scene = sc;
    ctx->have_scene = true;
    pc.lap("  scene: layouts (host) + the other hipMallocs and copies");
    rc = requantize(ctx, nullptr);
    pc.lap("  scene: fixed-point nodes on the device (first kernel launch: code object load)");
    return rc;
}

// tri_perm / up_indices (host copies of the mesh's orders) after a device-side install: fetched when a host-side path needs them
int refresh_host_mesh(rt_ctx *ctx) {
    if (!ctx->host_mesh_stale) return RT_OK;
    const size_t nt = (size_t)ctx->n_up_tris;
    std::vector<int4> up(nt);
    ctx->tri_perm.resize(nt);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipMemcpy(up.data(), ctx->tidx_up.p, nt * sizeof(int4), hipMemcpyDeviceToHost));
    RT_HIP(ctx, hipMemcpy(ctx->tri_perm.data(), ctx->perm_dev.p, nt * sizeof(int), hipMemcpyDeviceToHost));
    ctx->up_indices.resize(nt * 3);
    for (size_t t = 0; t < nt; ++t) { ctx->up_indices[3 * t] = up[t].x; ctx->up_indices[3 * t + 1] = up[t].y; ctx->up_indices[3 * t + 2] = up[t].z; }
    ctx->host_mesh_stale = false;
    return RT_OK;
}

// Several TriangleMesh objects in one scene (cpu:538-564): ONE tree for the traversal kernels.  Every mesh keeps the tree its own buildBVH made; the roots hang below synthetic
// internal nodes whose boxes are the unions of their children (exact: min / max of floats).  What this preserves:
//   * a mesh's triangles are tested iff the reference's own walk of that mesh reaches their leaf: the reference enters a mesh iff its root box is hit (cpu:279) and the
//     synthetic nodes above a root are entered whenever any root below them is -- BoundingBox::intersect is monotone along nested boxes: per axis the two plane parameters of
//     the larger box bracket the smaller box's (one rounding each of a monotone expression), an axis with u = 0 constrains neither box or both alike (the origin lies strictly inside
//     both intervals or the smaller box is missed), and a NaN on the first axis makes the smaller box a miss already;
//   * the synthetic tree is shaped so that the traversal order (right child first, cpu:291-292) reaches the meshes in OBJECT order, hence the visit-order triangle array holds them
//     mesh after mesh and min over (t, triangle index) = min over (t, object position, scan rank): the winner of the reference's loop over the objects with its strict '<' (cpu:554).
// real[k]: index into `meshes` of the k-th mesh with triangles, in object order.  Fills the combined arrays and f.m (which points into them).
struct Forest {
    std::vector<float> verts, arr;
    std::vector<int32_t> idx;
    std::vector<int> tri_off;                                           // per real mesh: first triangle in idx (+ the total at the end)
    rt_mesh m{};
};
int build_forest(rt_ctx *ctx, const rt_mesh *meshes, const std::vector<int> &real, Forest &f) {
    const int K = (int)real.size();
    std::vector<int> voff(K + 1, 0), noff(K + 1, 0);
    f.tri_off.assign(K + 1, 0);
    int64_t nv = 0, nt = 0, nn = K - 1;                                 // K - 1 synthetic nodes come first (node 0 = the forest's root)
    for (int k = 0; k < K; ++k) {
        const rt_mesh &m = meshes[real[k]];
        if (m.n_vertices < 0 || m.n_triangles < 0 || m.n_nodes < 0 || m.index_stride < 3) return fail(ctx, RT_ERR_INVALID, "mesh %d: bad sizes", real[k]);
        if ((m.n_vertices && !m.vertices) || (m.n_triangles && !m.indices) || (m.n_nodes && !m.bvh_arr10)) return fail(ctx, RT_ERR_INVALID, "mesh %d: array pointer is NULL", real[k]);
        voff[k] = (int)nv; f.tri_off[k] = (int)nt; noff[k] = (int)nn;
        nv += m.n_vertices; nt += m.n_triangles; nn += m.n_nodes;
        if (nv >= ((int64_t)1 << 31) || nt >= ((int64_t)1 << 31) || nn >= (1 << 24)) return fail(ctx, RT_ERR_INVALID, "the meshes together are too large (2^31 vertices / triangles, 2^24 nodes)");
    }
    voff[K] = (int)nv; f.tri_off[K] = (int)nt; noff[K] = (int)nn;
    f.verts.resize((size_t)nv * 3); f.idx.resize((size_t)nt * 3); f.arr.assign((size_t)nn * 10, 0.f);
    for (int k = 0; k < K; ++k) {
        const rt_mesh &m = meshes[real[k]];
        std::copy(m.vertices, m.vertices + (size_t)m.n_vertices * 3, f.verts.begin() + (size_t)voff[k] * 3);
        for (int t = 0; t < m.n_triangles; ++t)
            for (int c = 0; c < 3; ++c) {
                const int32_t v = m.indices[(size_t)t * m.index_stride + c];
                if (v < 0 || v >= m.n_vertices) return fail(ctx, RT_ERR_INVALID, "mesh %d: triangle %d references vertex %d outside [0,%d)", real[k], t, v, m.n_vertices);
                f.idx[3 * ((size_t)f.tri_off[k] + t) + c] = v + voff[k];
            }
        for (int n = 0; n < m.n_nodes; ++n) {
            const float *a = m.bvh_arr10 + (size_t)n * 10;
            float *o = f.arr.data() + ((size_t)noff[k] + n) * 10;
            const int l = (int)a[0], r = (int)a[1];
            if ((l != -1 && (l < 0 || l >= m.n_nodes)) || (r != -1 && (r < 0 || r >= m.n_nodes))) return fail(ctx, RT_ERR_INVALID, "mesh %d: bvh_arr10 node %d has a child index out of range", real[k], n);
            const int ts = (int)a[8], te = (int)a[9];
            if (ts < 0 || te < ts || te > m.n_triangles) return fail(ctx, RT_ERR_INVALID, "mesh %d: bvh_arr10 node %d has triangle range [%d,%d) outside [0,%d)", real[k], n, ts, te, m.n_triangles);
            o[0] = l == -1 ? -1.f : (float)(l + noff[k]); o[1] = r == -1 ? -1.f : (float)(r + noff[k]);
            for (int c = 2; c < 8; ++c) o[c] = a[c];
            o[8] = (float)(ts + f.tri_off[k]); o[9] = (float)(te + f.tri_off[k]);
        }
    }
    // the synthetic nodes: meshes [a, b) below node `self`; the RIGHT child holds the first half (visited first)
    int next_syn = 1;
    struct Job { int a, b, self; };
    std::vector<Job> jobs{{0, K, 0}};
    std::vector<Job> post;
    while (!jobs.empty()) {
        const Job j = jobs.back(); jobs.pop_back();
        post.push_back(j);
        const int mid = j.a + (j.b - j.a + 1) / 2;
        auto child = [&](int a, int b) { if (b - a == 1) return noff[a]; const int id = next_syn++; jobs.push_back({a, b, id}); return id; };
        float *o = f.arr.data() + (size_t)j.self * 10;
        o[1] = (float)child(j.a, mid);                                  // right = the earlier meshes
        o[0] = (float)child(mid, j.b);
        o[8] = (float)f.tri_off[j.a]; o[9] = (float)f.tri_off[j.b];
    }
    for (size_t q = post.size(); q-- > 0;) {                            // children before parents: a synthetic node's index is larger than its parent's
        float *o = f.arr.data() + (size_t)post[q].self * 10;
        const float *l = f.arr.data() + (size_t)(int)o[0] * 10, *r = f.arr.data() + (size_t)(int)o[1] * 10;
        for (int c = 0; c < 3; ++c) { o[2 + c] = std::min(l[2 + c], r[2 + c]); o[5 + c] = std::max(l[5 + c], r[5 + c]); }
    }
    f.m = rt_mesh{};
    f.m.vertices = f.verts.data(); f.m.n_vertices = (int)nv; f.m.indices = f.idx.data(); f.m.index_stride = 3; f.m.n_triangles = (int)nt;
    f.m.bvh_arr10 = f.arr.data(); f.m.n_nodes = (int)nn;
    f.m.object_slot = meshes[real[0]].object_slot;
    return RT_OK;
}

}  // namespace
