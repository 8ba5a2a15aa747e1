// rt_host_mesh.hip.h -- host side, part 4 of 4 (inside rt_capi.hip's extern "C" block): the entry points that change the uploaded mesh on the device -- smooth normals,
// transform + refit, rebuild of the reference's tree, the LBVH builder and its device-side install.
#pragma once

int rt_mesh_set_normals(rt_ctx *ctx, const float *normals_xyz, int n_normals, const int32_t *nidx, int index_stride, int n_triangles) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_OWN_STREAM(ctx);
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_scene_upload has not been called");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(own_stream(ctx)));
    if (!normals_xyz || !nidx) { ctx->scene.nrm = nullptr; return RT_OK; }          // back to flat shading
    if (ctx->n_real_meshes > 1) return fail(ctx, RT_ERR_UNSUPPORTED, "the scene holds %d meshes: smooth normals are set for ONE TriangleMesh", ctx->n_real_meshes);
    if (int rr = refresh_host_mesh(ctx); rr != RT_OK) return rr;
    if (ctx->scene.mesh_slot < 0) return fail(ctx, RT_ERR_INVALID, "the scene has no mesh");
    if (n_normals <= 0 || index_stride < 3) return fail(ctx, RT_ERR_INVALID, "bad normal array sizes");
    std::vector<float4> nr(ctx->tri_perm.size() * 3);
    for (size_t t = 0; t < ctx->tri_perm.size(); ++t) {
        const int src = ctx->tri_perm[t];
        if (src < 0 || src >= n_triangles) return fail(ctx, RT_ERR_INVALID, "n_triangles %d does not cover the uploaded mesh", n_triangles);
        for (int k = 0; k < 3; ++k) {
            const int ni = nidx[(size_t)src * index_stride + k];
            if (ni < 0 || ni >= n_normals) return fail(ctx, RT_ERR_INVALID, "triangle %d references normal %d outside [0,%d)", src, ni, n_normals);
            nr[3 * t + k] = make_float4(normals_xyz[3 * (size_t)ni], normals_xyz[3 * (size_t)ni + 1], normals_xyz[3 * (size_t)ni + 2], 0.f);
        }
    }
    int rc = upload(ctx, ctx->nrm, nr.data(), nr.size() * sizeof(float4));
    if (rc != RT_OK) return rc;
    ctx->scene.nrm = static_cast<const float4 *>(ctx->nrm.p);
    return RT_OK;
}

int rt_mesh_transform(rt_ctx *ctx, const float rotation[9], const float translation[3]) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_OWN_STREAM(ctx);
    if (!rotation || !translation) return fail(ctx, RT_ERR_INVALID, "rotation/translation is NULL");
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_scene_upload has not been called");
    rtk::Scene &sc = ctx->scene;
    if (sc.mesh_slot < 0 || sc.n_nodes <= 0 || sc.n_verts <= 0) return RT_OK;     // no mesh: nothing to move
    RT_HIP(ctx, hipSetDevice(ctx->device));
    rtk::Mat3 m;
    for (int k = 0; k < 9; ++k) m.r[k] = rotation[k];
    for (int k = 0; k < 3; ++k) m.t[k] = translation[k];
    hipLaunchKernelGGL(rtk::transform_kernel, dim3((unsigned)((sc.n_verts + 255) / 256)), dim3(256), 0, own_stream(ctx),
                       static_cast<float4 *>(ctx->verts.p), sc.n_verts, m);
    if (sc.nrm != nullptr)      // the reference's kernel rotates the normals and ADDS the translation to them as well (global_launcher.cu:357-363)
        hipLaunchKernelGGL(rtk::transform_kernel, dim3((unsigned)((3 * sc.n_tris + 255) / 256)), dim3(256), 0, own_stream(ctx),
                           static_cast<float4 *>(ctx->nrm.p), 3 * sc.n_tris, m);
    hipLaunchKernelGGL(rtk::retri_kernel, dim3((unsigned)((sc.n_tris + 255) / 256)), dim3(256), 0, own_stream(ctx),
                       static_cast<const int4 *>(ctx->tidx.p), static_cast<const float4 *>(ctx->verts.p), static_cast<float4 *>(ctx->tri.p), sc.n_tris);
    rtk::RefitArgs a{};
    a.node_lo = static_cast<float4 *>(ctx->node_lo.p); a.node_hi = static_cast<float4 *>(ctx->node_hi.p);
    a.nodes2 = static_cast<float4 *>(ctx->nodes2.p); a.nodesq = static_cast<float4 *>(ctx->nodesq.p); a.nodesb = static_cast<float4 *>(ctx->nodesb.p);
    a.q2thr = static_cast<const int *>(ctx->q2thr.p); a.left_of = static_cast<const int *>(ctx->left_dev.p);
    a.lvl_nodes = static_cast<const int *>(ctx->lvl_nodes.p); a.lvl_off = static_cast<const int *>(ctx->lvl_off.p);
    a.tidx = static_cast<const int4 *>(ctx->tidx.p); a.verts = static_cast<const float4 *>(ctx->verts.p);
    a.n_nodes = sc.n_nodes; a.n_levels = ctx->n_levels;
    hipLaunchKernelGGL(rtk::refit_kernel, dim3(1), dim3(1024), 0, own_stream(ctx), a);
    RT_HIP(ctx, hipGetLastError());
    // the root box travels as a kernel argument (uniform root-box pre-test): fetch the refitted one
    float4 root[2];
    RT_HIP(ctx, hipMemcpyAsync(&root[0], ctx->node_lo.p, sizeof(float4), hipMemcpyDeviceToHost, own_stream(ctx)));
    RT_HIP(ctx, hipMemcpyAsync(&root[1], ctx->node_hi.p, sizeof(float4), hipMemcpyDeviceToHost, own_stream(ctx)));
    RT_HIP(ctx, hipStreamSynchronize(own_stream(ctx)));
    sc.root_lo = root[0]; sc.root_hi = root[1];
    // the refitted root box contains every node's (unions, bottom-up): it bounds the magnitudes wf_travq's box filter needs
    const float rv[6] = {root[0].x, root[0].y, root[0].z, root[1].x, root[1].y, root[1].z};
    bool fast = true;
    float bm[3];
    for (int a = 0; a < 3; ++a) {
        if (!(rv[a] <= rv[a + 3]) || !(std::fabs(rv[a]) < 1e8f) || !(std::fabs(rv[a + 3]) < 1e8f)) fast = false;
        bm[a] = std::max(std::fabs(rv[a]), std::fabs(rv[a + 3]));
    }
    sc.bmx = bm[0]; sc.bmy = bm[1]; sc.bmz = bm[2];
    sc.fast_box = fast ? 1 : 0;
    return requantize(ctx, own_stream(ctx));                                          // the fixed-point pairs follow the refitted boxes (same topology: q16_topo_ok stands; unions nest)
}

// TriangleMesh::buildBVH on the device, bit for bit (rt_bvhbuild.hip.h): leaves the flat tree in ctx->bb_arr and the triangle order in ctx->bb_idx
static int rebuild_reference_tree(rt_ctx *ctx, const int nt, int &n_nodes_out) {
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cap = 2 * (size_t)nt + 2;                                          // nodes: every split makes two
    int rc;
    if ((rc = ensure(ctx, ctx->bb_idx, nt * sizeof(int))) != RT_OK || (rc = ensure(ctx, ctx->bb_cnt, nt * sizeof(int))) != RT_OK ||
        (rc = ensure(ctx, ctx->bb_pa, nt * sizeof(int))) != RT_OK || (rc = ensure(ctx, ctx->bb_pb, nt * sizeof(int))) != RT_OK ||
        (rc = ensure(ctx, ctx->bb_tmp, nt * sizeof(int))) != RT_OK || (rc = ensure(ctx, ctx->bb_nodes_i, 4 * cap * sizeof(int))) != RT_OK ||
        (rc = ensure(ctx, ctx->bb_nodes_f, 2 * cap * sizeof(float4))) != RT_OK || (rc = ensure(ctx, ctx->bb_counter, 2 * sizeof(int))) != RT_OK ||
        (rc = ensure(ctx, ctx->bb_lvl, (cap + 1) * sizeof(int))) != RT_OK || (rc = ensure(ctx, ctx->bb_size, cap * sizeof(int))) != RT_OK ||
        (rc = ensure(ctx, ctx->bb_pre, cap * sizeof(int))) != RT_OK || (rc = ensure(ctx, ctx->bb_arr, cap * 10 * sizeof(float))) != RT_OK)
        return rc;
    rtk::BuildArgs a{};
    a.verts = static_cast<const float4 *>(ctx->verts.p); a.tidx_up = static_cast<const int4 *>(ctx->tidx_up.p);
    a.idx = static_cast<int *>(ctx->bb_idx.p); a.cnt = static_cast<int *>(ctx->bb_cnt.p);
    a.ptr_a = static_cast<int *>(ctx->bb_pa.p); a.ptr_b = static_cast<int *>(ctx->bb_pb.p); a.tmp = static_cast<int *>(ctx->bb_tmp.p);
    int *ni = static_cast<int *>(ctx->bb_nodes_i.p);
    a.n_start = ni; a.n_end = ni + cap; a.n_left = ni + 2 * cap; a.n_right = ni + 3 * cap;
    a.n_mn = static_cast<float4 *>(ctx->bb_nodes_f.p); a.n_mx = a.n_mn + cap;
    a.counter = static_cast<int *>(ctx->bb_counter.p); a.n_tris = nt; a.cap = (int)cap;
    hipStream_t q = own_stream(ctx);
    // root = node 0 over all triangles (buildBVH(&bvh, 0, T), cpu:684); the permutation starts as the identity
    hipLaunchKernelGGL(rtk::iota_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, q, a.idx, nt);
    const int root_range[2] = {0, nt}, one[2] = {1, 0};              // counter[0] = nodes allocated, counter[1] = a split was refused for lack of capacity
    RT_HIP(ctx, hipMemcpyAsync(a.n_start, &root_range[0], sizeof(int), hipMemcpyHostToDevice, q));
    RT_HIP(ctx, hipMemcpyAsync(a.n_end, &root_range[1], sizeof(int), hipMemcpyHostToDevice, q));
    RT_HIP(ctx, hipMemcpyAsync(a.counter, one, 2 * sizeof(int), hipMemcpyHostToDevice, q));
    RT_HIP(ctx, hipStreamSynchronize(q));                                           // the three sources above live on this stack frame
    std::vector<int> lvl_first{0};
    int first = 0, count = 1;
    while (count > 0) {                                                             // one launch per level, one workgroup per node
        hipLaunchKernelGGL(rtk::bvh_level_kernel, dim3((unsigned)count), dim3(rtk::kBuildThreads), 0, q, a, first);
        RT_HIP(ctx, hipGetLastError());
        int tot[2] = {0, 0};
        RT_HIP(ctx, hipMemcpyAsync(tot, a.counter, 2 * sizeof(int), hipMemcpyDeviceToHost, q));
        RT_HIP(ctx, hipStreamSynchronize(q));
        const int total = tot[0];
        // the kernel refuses a split that would pass the arrays' capacity (it cannot for a tree over nt triangles); the scene in use is untouched so far
        if (tot[1] != 0 || (size_t)total > cap) return fail(ctx, RT_ERR_INTERNAL, "BVH build needed more than %zu nodes for %d triangles (scene unchanged)", cap, nt);
        first += count;
        lvl_first.push_back(first);
        count = total - first;
    }
    const int n_nodes = first, n_levels = (int)lvl_first.size() - 1;
    n_nodes_out = n_nodes;
    if (n_nodes >= (1 << 24)) return fail(ctx, RT_ERR_INVALID, "node indices are stored as floats: < 2^24 nodes");
    RT_HIP(ctx, hipMemcpyAsync(ctx->bb_lvl.p, lvl_first.data(), lvl_first.size() * sizeof(int), hipMemcpyHostToDevice, q));
    hipLaunchKernelGGL(rtk::bvh_flatten_kernel, dim3(1), dim3(1024), 0, q, a, static_cast<const int *>(ctx->bb_lvl.p), n_levels,
                       static_cast<int *>(ctx->bb_size.p), static_cast<int *>(ctx->bb_pre.p), static_cast<float *>(ctx->bb_arr.p));
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

// The LBVH builder (rt_lbvh.hip.h): Morton sort + parallel hierarchy emission, leaves cut by the surface-area heuristic (at most kLbvhLeaf = 32 triangles); same outputs
static int rebuild_lbvh_tree(rt_ctx *ctx, const int nt, int &n_nodes_out) {
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t q = own_stream(ctx);
    const size_t n = (size_t)nt, nc = 2 * n - 1;
    int rc;
    DevBuf &B = ctx->lb_pool;
    // one pool, carved: keys (2 x 8n), vals (2 x 4n), 6 int arrays of n, flags, boxes (4 x 16n), alive + index (2 x 4 (2n)), bounds / stats
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_keys = carve(8 * n), o_keys2 = carve(8 * n), o_vals = carve(4 * n), o_vals2 = carve(4 * n);
    const size_t o_left = carve(4 * n), o_right = carve(4 * n), o_parent = carve(4 * n), o_first = carve(4 * n), o_last = carve(4 * n), o_lparent = carve(4 * n), o_flag = carve(4 * n), o_cost = carve(4 * n), o_leafify = carve(4 * n);
    const size_t o_ilo = carve(16 * n), o_ihi = carve(16 * n), o_llo = carve(16 * n), o_lhi = carve(16 * n);
    const size_t o_alive = carve(4 * nc), o_index = carve(4 * nc), o_small = carve(64);
    size_t sort_tmp = 0, scan_tmp = 0;
    {
        unsigned long long *k0 = nullptr; int *v0 = nullptr;
        if (rocprim::radix_sort_pairs(nullptr, sort_tmp, k0, k0, v0, v0, n, 0, 63, q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::radix_sort_pairs (size query) failed");
        if (rocprim::exclusive_scan(nullptr, scan_tmp, v0, v0, 0, nc, rocprim::plus<int>(), q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::exclusive_scan (size query) failed");
    }
    const size_t o_tmp = carve(std::max(sort_tmp, scan_tmp) + 256);
    if ((rc = ensure(ctx, B, off)) != RT_OK) return rc;
    if ((rc = ensure(ctx, ctx->bb_idx, n * sizeof(int))) != RT_OK) return rc;
    uint8_t *base = static_cast<uint8_t *>(B.p);
    rtk::LbvhArgs a{};
    // the leaf cut's triangle cost: kLbvhCt (wf_travq's step times on 64-byte pairs) for small trees; 1.0 for trees that will use the 32-byte fixed-point pairs, where a box test is
    // cheaper still but a triangle's 48-byte gather is not (swept on 524 288 / 2 M triangles: Ct 1.0 / 1.6 / 2.5 / 4 / 8 = 2.77 / 2.84 / 2.99 / 3.04 / 3.06 and 8.21 / 8.40 / 8.72 / 8.73 / 8.80 ms per frame)
    a.ct = ctx->knobs.lbvh_ct > 0.f ? ctx->knobs.lbvh_ct : (nt >= kQ16AutoNodes ? 1.0f : rtk::kLbvhCt); a.cb = rtk::kLbvhCb;
    a.verts = static_cast<const float4 *>(ctx->verts.p); a.tidx_up = static_cast<const int4 *>(ctx->tidx_up.p); a.n = nt;
    a.bounds = reinterpret_cast<unsigned int *>(base + o_small); a.stats = reinterpret_cast<int *>(base + o_small + 32);
    a.keys = reinterpret_cast<unsigned long long *>(base + o_keys); a.vals = reinterpret_cast<int *>(base + o_vals);
    a.left = reinterpret_cast<int *>(base + o_left); a.right = reinterpret_cast<int *>(base + o_right); a.parent = reinterpret_cast<int *>(base + o_parent);
    a.first = reinterpret_cast<int *>(base + o_first); a.last = reinterpret_cast<int *>(base + o_last); a.leaf_parent = reinterpret_cast<int *>(base + o_lparent);
    a.flag = reinterpret_cast<int *>(base + o_flag);
    a.cost = reinterpret_cast<float *>(base + o_cost); a.leafify = reinterpret_cast<int *>(base + o_leafify);
    a.ibox_lo = reinterpret_cast<float4 *>(base + o_ilo); a.ibox_hi = reinterpret_cast<float4 *>(base + o_ihi);
    a.lbox_lo = reinterpret_cast<float4 *>(base + o_llo); a.lbox_hi = reinterpret_cast<float4 *>(base + o_lhi);
    a.alive = reinterpret_cast<int *>(base + o_alive); a.index = reinterpret_cast<int *>(base + o_index);
    const unsigned int binit[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    int zero4[4] = {0, 0, 0, 0};
    RT_HIP(ctx, hipMemcpyAsync(a.bounds, binit, sizeof(binit), hipMemcpyHostToDevice, q));
    RT_HIP(ctx, hipMemcpyAsync(a.stats, zero4, sizeof(zero4), hipMemcpyHostToDevice, q));
    RT_HIP(ctx, hipStreamSynchronize(q));                                        // (the two sources live on this stack frame)
    const dim3 gt((unsigned)((n + 255) / 256)), gc((unsigned)((nc + 255) / 256)), blk(256);
    hipLaunchKernelGGL(rtk::lbvh_bounds_kernel, gt, blk, 0, q, a);
    hipLaunchKernelGGL(rtk::lbvh_morton_kernel, gt, blk, 0, q, a);
    {   // (code, triangle) pairs by code; the sorted arrays become a.keys / a.vals
        unsigned long long *k2 = reinterpret_cast<unsigned long long *>(base + o_keys2);
        int *v2 = reinterpret_cast<int *>(base + o_vals2);
        size_t tmp = sort_tmp;
        if (rocprim::radix_sort_pairs(base + o_tmp, tmp, a.keys, k2, a.vals, v2, n, 0, 63, q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::radix_sort_pairs failed");
        a.keys = k2; a.vals = v2;
    }
    hipLaunchKernelGGL(rtk::lbvh_hierarchy_kernel, gt, blk, 0, q, a);
    hipLaunchKernelGGL(rtk::lbvh_boxes_kernel, gt, blk, 0, q, a);          // boxes + the leaf-or-subtree decision, bottom-up
    hipLaunchKernelGGL(rtk::lbvh_alive_kernel, gc, blk, 0, q, a);
    {
        size_t tmp = scan_tmp;
        if (rocprim::exclusive_scan(base + o_tmp, tmp, a.alive, a.index, 0, nc, rocprim::plus<int>(), q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::exclusive_scan failed");
    }
    RT_HIP(ctx, hipGetLastError());
    int last2[2] = {0, 0};                                                       // n_alive = index[last] + alive[last]
    RT_HIP(ctx, hipMemcpyAsync(&last2[0], a.index + (nc - 1), sizeof(int), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipMemcpyAsync(&last2[1], a.alive + (nc - 1), sizeof(int), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipStreamSynchronize(q));
    const int n_nodes = last2[0] + last2[1];
    if (n_nodes < 1 || (size_t)n_nodes > nc) return fail(ctx, RT_ERR_INTERNAL, "LBVH build: %d nodes for %d triangles (scene unchanged)", n_nodes, nt);
    if (n_nodes >= (1 << 24)) return fail(ctx, RT_ERR_INVALID, "node indices are stored as floats: < 2^24 nodes");
    if ((rc = ensure(ctx, ctx->bb_arr, (size_t)n_nodes * 10 * sizeof(float))) != RT_OK) return rc;
    a.arr10 = static_cast<float *>(ctx->bb_arr.p);
    hipLaunchKernelGGL(rtk::lbvh_emit_kernel, gc, blk, 0, q, a);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(ctx->bb_idx.p, a.vals, n * sizeof(int), hipMemcpyDeviceToDevice, q));   // the triangle order, where the shared tail expects it
    int st[4] = {0, 0, 0, 0};
    RT_HIP(ctx, hipMemcpyAsync(st, a.stats, sizeof(st), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipStreamSynchronize(q));
    ctx->build.n_leaves = st[0]; ctx->build.max_leaf_tris = st[1]; ctx->build.max_depth = st[2];
    ctx->lb_args = a;
    n_nodes_out = n_nodes;
    return RT_OK;
}

// The render kernels' formats from the LBVH builder's arrays, on the device (rt_lbvh.hip.h, second half): what install_scene does on the
// host for an uploaded tree.  `old`: the scene in use (spheres, light, camera, albedo, mesh slot carry over).
static int install_lbvh_device(rt_ctx *ctx, const rtk::Scene &old, const int n_nodes) {
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t q = own_stream(ctx);
    const rtk::LbvhArgs &a = ctx->lb_args;
    const size_t n = (size_t)a.n, N = (size_t)n_nodes, nc = 2 * n - 1;
    int rc;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t o_flag = carve(4 * (n + 1)), o_scan = carve(4 * (n + 1)), o_X = carve(4 * N), o_bfs = carve(4 * N), o_key = carve(8 * N), o_key2 = carve(8 * N),
                 o_val = carve(4 * N), o_val2 = carve(4 * N), o_hist = carve(4 * 80), o_upnew = carve(16 * n);
    size_t sort_tmp = 0, scan_tmp = 0;
    {
        unsigned long long *k0 = nullptr; int *v0 = nullptr;
        if (rocprim::radix_sort_pairs(nullptr, sort_tmp, k0, k0, v0, v0, N, 0, 64, q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::radix_sort_pairs (size query) failed");
        if (rocprim::exclusive_scan(nullptr, scan_tmp, v0, v0, 0, n + 1, rocprim::plus<int>(), q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::exclusive_scan (size query) failed");
    }
    const size_t o_tmp = carve(std::max(sort_tmp, scan_tmp) + 256);
    if ((rc = ensure(ctx, ctx->lb_pool2, off)) != RT_OK) return rc;
    // the scene in use stays untouched until every allocation has succeeded
    DevBuf *outs[] = {&ctx->node_lo, &ctx->node_hi, &ctx->nodes2, &ctx->nodesq, &ctx->nodesb, &ctx->q2thr, &ctx->left_dev, &ctx->lvl_nodes, &ctx->lvl_off, &ctx->tri, &ctx->tidx, &ctx->perm_dev};
    const size_t need[] = {N * 16, N * 16, 2 * N * 16, 2 * (N + 1) * 16, 2 * (N + 1) * 16, (N + 1) * 4, N * 4, N * 4, 80 * 4, 3 * n * 16, n * 16, n * 4};
    ctx->have_scene = false;                                                     // (a failure from here on leaves the context without a scene, as the host path does)
    for (size_t k = 0; k < sizeof(need) / sizeof(need[0]); ++k) if ((rc = ensure(ctx, *outs[k], need[k])) != RT_OK) return rc;
    uint8_t *base = static_cast<uint8_t *>(ctx->lb_pool2.p);
    int *flag = reinterpret_cast<int *>(base + o_flag);
    rtk::LbvhLayout y{};
    y.n_nodes = n_nodes;
    y.lscan = reinterpret_cast<int *>(base + o_scan); y.X = reinterpret_cast<int *>(base + o_X); y.bfs = reinterpret_cast<int *>(base + o_bfs);
    y.bkey = reinterpret_cast<unsigned long long *>(base + o_key); y.bval = reinterpret_cast<int *>(base + o_val);
    y.dhist = reinterpret_cast<int *>(base + o_hist);
    y.node_lo = static_cast<float4 *>(ctx->node_lo.p); y.node_hi = static_cast<float4 *>(ctx->node_hi.p); y.nodes2 = static_cast<float4 *>(ctx->nodes2.p);
    y.nodesq = static_cast<float4 *>(ctx->nodesq.p); y.nodesb = static_cast<float4 *>(ctx->nodesb.p); y.q2thr = static_cast<int *>(ctx->q2thr.p);
    y.left_of = static_cast<int *>(ctx->left_dev.p); y.lvl_nodes = static_cast<int *>(ctx->lvl_nodes.p);
    y.tidx_visit = static_cast<int4 *>(ctx->tidx.p); y.tidx_up_new = reinterpret_cast<int4 *>(base + o_upnew); y.perm = static_cast<int *>(ctx->perm_dev.p);
    RT_HIP(ctx, hipMemsetAsync(flag, 0, 4 * (n + 1), q));
    RT_HIP(ctx, hipMemsetAsync(y.dhist, 0, 4 * 80, q));
    RT_HIP(ctx, hipMemsetAsync(y.nodesq, 0, 32, q));                              // entry 0 of the breadth-first arrays is padding
    RT_HIP(ctx, hipMemsetAsync(y.nodesb, 0, 32, q));
    RT_HIP(ctx, hipMemsetAsync(y.q2thr, 0, 4, q));
    const dim3 gt((unsigned)((n + 255) / 256)), gc((unsigned)((nc + 255) / 256)), gn((unsigned)((N + 255) / 256)), blk(256);
    hipLaunchKernelGGL(rtk::lbvh_leafflag_kernel, gc, blk, 0, q, a, flag);
    { size_t tmp = scan_tmp; if (rocprim::exclusive_scan(base + o_tmp, tmp, flag, y.lscan, 0, n + 1, rocprim::plus<int>(), q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::exclusive_scan failed"); }
    hipLaunchKernelGGL(rtk::lbvh_walk_kernel, gc, blk, 0, q, a, y);
    unsigned long long *key2 = reinterpret_cast<unsigned long long *>(base + o_key2);
    int *val2 = reinterpret_cast<int *>(base + o_val2);
    { size_t tmp = sort_tmp; if (rocprim::radix_sort_pairs(base + o_tmp, tmp, y.bkey, key2, y.bval, val2, N, 0, 64, q) != hipSuccess) return fail(ctx, RT_ERR_HIP, "rocprim::radix_sort_pairs failed"); }
    hipLaunchKernelGGL(rtk::lbvh_rank_kernel, gn, blk, 0, q, y, val2);
    const int4 *up_old = static_cast<const int4 *>(ctx->tidx_up.p);
    hipLaunchKernelGGL(rtk::lbvh_layout_kernel, gc, blk, 0, q, a, y, up_old);
    hipLaunchKernelGGL(rtk::lbvh_reorder_kernel, gt, blk, 0, q, a, up_old, y.tidx_up_new);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(ctx->tidx_up.p, y.tidx_up_new, n * sizeof(int4), hipMemcpyDeviceToDevice, q));   // the sorted order is the new uploaded order
    hipLaunchKernelGGL(rtk::retri_kernel, gt, blk, 0, q, static_cast<const int4 *>(ctx->tidx.p), static_cast<const float4 *>(ctx->verts.p), static_cast<float4 *>(ctx->tri.p), (int)n);
    RT_HIP(ctx, hipGetLastError());
    int hist[65];
    float4 root[2];
    RT_HIP(ctx, hipMemcpyAsync(hist, y.dhist, sizeof(hist), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipMemcpyAsync(&root[0], ctx->node_lo.p, sizeof(float4), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipMemcpyAsync(&root[1], ctx->node_hi.p, sizeof(float4), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipStreamSynchronize(q));
    const int maxd = hist[64];
    if (maxd > 58) return fail(ctx, RT_ERR_INTERNAL, "LBVH layout: depth %d exceeds the 58 path bits of the breadth-first sort key", maxd);
    std::vector<int> lvl_off(maxd + 2, 0);
    for (int d = 0; d <= maxd; ++d) lvl_off[d + 1] = lvl_off[d] + hist[d];
    if (lvl_off[maxd + 1] != n_nodes) return fail(ctx, RT_ERR_INTERNAL, "LBVH layout: %d nodes in the depth histogram, %d in the tree", lvl_off[maxd + 1], n_nodes);
    if ((rc = upload(ctx, ctx->lvl_off, lvl_off.data(), lvl_off.size() * sizeof(int))) != RT_OK) return rc;
    ctx->n_levels = maxd + 1;
    rtk::Scene sc = old;
    sc.nrm = nullptr;
    sc.n_nodes = n_nodes; sc.n_tris = (int)n;
    mesh_table_single(sc, ctx->real_obj);
    sc.root_lo = root[0]; sc.root_hi = root[1];
    bool fast = true;
    const float rv[6] = {root[0].x, root[0].y, root[0].z, root[1].x, root[1].y, root[1].z};
    float bm[3];
    for (int k = 0; k < 3; ++k) {                                                // every box nests inside the root's (unions, bottom-up), min <= max by construction
        if (!(rv[k] <= rv[k + 3]) || !(std::fabs(rv[k]) < 1e8f) || !(std::fabs(rv[k + 3]) < 1e8f)) fast = false;
        bm[k] = std::max(std::fabs(rv[k]), std::fabs(rv[k + 3]));
    }
    sc.bmx = bm[0]; sc.bmy = bm[1]; sc.bmz = bm[2]; sc.fast_box = fast ? 1 : 0;
    sc.node_lo = static_cast<const float4 *>(ctx->node_lo.p); sc.node_hi = static_cast<const float4 *>(ctx->node_hi.p);
    sc.nodes = static_cast<const float4 *>(ctx->nodes2.p); sc.nodesq = static_cast<const float4 *>(ctx->nodesq.p); sc.nodesb = static_cast<const float4 *>(ctx->nodesb.p);
    sc.q2thr = static_cast<const int *>(ctx->q2thr.p); sc.tri = static_cast<const float4 *>(ctx->tri.p);
    sc.verts = static_cast<const float4 *>(ctx->verts.p); sc.tidx = static_cast<const int4 *>(ctx->tidx.p);
    ctx->travq_ok = n_nodes + 2 < (1 << rtk::kQNodeBits) && (uint64_t)n * 48 < ((uint64_t)1 << 32);   // leaves hold at most kLbvhLeaf triangles
    ctx->scene = sc;
    ctx->host_mesh_stale = true;                                                 // tri_perm / up_indices: on the device now (perm_dev, tidx_up)
    ctx->have_scene = true;
    ctx->q16_topo_ok = true;                                                     // boxes are unions, bottom-up: they nest
    ctx->qw_topo_ok = true;                                                      // (an LBVH leaf holds at least one triangle)
    ctx->q16_leaf_shift = rtk::q16_leaf_shift(rtk::kLbvhLeaf, n);                // leaves of at most kLbvhLeaf triangles
    return requantize(ctx, q);
}

int rt_mesh_rebuild_mode(rt_ctx *ctx, int mode, float *bvh_arr10_out, int32_t *tri_order_out, int32_t *n_nodes_out) {
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "ctx is NULL");
    RT_OWN_STREAM(ctx);
    if (mode != RT_BVH_REFERENCE && mode != RT_BVH_LBVH) return fail(ctx, RT_ERR_INVALID, "unknown BVH mode %d", mode);
    if (!ctx->have_scene) return fail(ctx, RT_ERR_NO_SCENE, "rt_scene_upload has not been called");
    if (n_nodes_out) *n_nodes_out = 0;
    const rtk::Scene old = ctx->scene;
    const int nt = ctx->n_up_tris, nv = old.n_verts;
    if (ctx->n_real_meshes > 1) return fail(ctx, RT_ERR_UNSUPPORTED, "the scene holds %d meshes: a rebuild works on ONE TriangleMesh (upload the rebuilt meshes again)", ctx->n_real_meshes);
    if (old.mesh_slot < 0 || ctx->real_obj < 0 || nt <= 0 || nv <= 0) return RT_OK;  // no mesh: nothing to build
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t q = own_stream(ctx);
    int rc;
    int n_nodes = 0;
    ctx->build = rt_build_stats{};
    hipEvent_t e0 = ctx->ev_t0, e1 = ctx->ev_t1;                                    // (the tone-mapping events are free here: nothing else runs on the stream)
    RT_HIP(ctx, hipEventRecord(e0, q));
    // a mesh of a single leaf's worth of triangles is a single leaf in either mode (cpu:217: fewer than five triangles are never split)
    if (mode == RT_BVH_LBVH && nt > 4) rc = rebuild_lbvh_tree(ctx, nt, n_nodes);
    else { mode = RT_BVH_REFERENCE; rc = rebuild_reference_tree(ctx, nt, n_nodes); }
    if (rc != RT_OK) return rc;
    RT_HIP(ctx, hipEventRecord(e1, q));
    RT_HIP(ctx, hipEventSynchronize(e1));
    RT_HIP(ctx, hipEventElapsedTime(&ctx->build.device_build_ms, e0, e1));
    ctx->have_tonemap_time = false;                                                 // (the borrowed events no longer bracket a tone mapping)
    ctx->build.mode = mode; ctx->build.n_nodes = n_nodes; ctx->build.n_triangles = nt;
    const auto t_install = std::chrono::steady_clock::now();
    int *const order_dev = static_cast<int *>(ctx->bb_idx.p);
    if (mode == RT_BVH_LBVH && old.nrm == nullptr && !ctx->lbvh_host_install && ctx->build.max_depth <= 56) {
        // the kernels' formats straight from the builder's arrays; the flat tree and the order travel to the host only if the caller asks
        if ((rc = install_lbvh_device(ctx, old, n_nodes)) != RT_OK) return rc;
        if (bvh_arr10_out) RT_HIP(ctx, hipMemcpyAsync(bvh_arr10_out, ctx->bb_arr.p, (size_t)n_nodes * 10 * sizeof(float), hipMemcpyDeviceToHost, q));
        if (tri_order_out) RT_HIP(ctx, hipMemcpyAsync(tri_order_out, order_dev, (size_t)nt * sizeof(int), hipMemcpyDeviceToHost, q));
        RT_HIP(ctx, hipStreamSynchronize(q));
        if (n_nodes_out) *n_nodes_out = n_nodes;
        ctx->build.install_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_install).count();
        ctx->build.install_on_device = 1;
        return RT_OK;
    }
    if ((rc = refresh_host_mesh(ctx)) != RT_OK) return rc;                          // the host path below starts from up_indices / tri_perm
    // The tree is built.  The O(n) re-layout for the kernels (traversal order, visit-order triangle records, sibling pairs, refit
    // levels) reuses the upload path on the host: ~30 bytes per triangle over PCIe each way.
    std::vector<float> arr((size_t)n_nodes * 10);
    std::vector<int> order(nt);
    std::vector<float4> hv(nv);
    RT_HIP(ctx, hipMemcpyAsync(arr.data(), ctx->bb_arr.p, arr.size() * sizeof(float), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipMemcpyAsync(order.data(), order_dev, order.size() * sizeof(int), hipMemcpyDeviceToHost, q));
    RT_HIP(ctx, hipMemcpyAsync(hv.data(), ctx->verts.p, hv.size() * sizeof(float4), hipMemcpyDeviceToHost, q));
    std::vector<float4> old_nrm;
    if (old.nrm != nullptr) {
        old_nrm.resize((size_t)old.n_tris * 3);
        RT_HIP(ctx, hipMemcpyAsync(old_nrm.data(), ctx->nrm.p, old_nrm.size() * sizeof(float4), hipMemcpyDeviceToHost, q));
    }
    RT_HIP(ctx, hipStreamSynchronize(q));
    std::vector<float> vx((size_t)nv * 3);
    for (int i = 0; i < nv; ++i) { vx[3 * (size_t)i] = hv[i].x; vx[3 * (size_t)i + 1] = hv[i].y; vx[3 * (size_t)i + 2] = hv[i].z; }
    std::vector<int32_t> ix((size_t)nt * 3);
    for (int t = 0; t < nt; ++t) for (int k = 0; k < 3; ++k) ix[3 * (size_t)t + k] = ctx->up_indices[3 * (size_t)order[t] + k];
    const std::vector<int> old_perm = ctx->tri_perm;                               // old visit order -> old uploaded order
    rt_mesh m{};
    m.vertices = vx.data(); m.n_vertices = nv; m.indices = ix.data(); m.index_stride = 3; m.n_triangles = nt;
    m.bvh_arr10 = arr.data(); m.n_nodes = n_nodes;
    m.object_slot = ctx->real_obj;                                                  // (albedo and material stay in the scene's mesh table, which `sc` carries over)
    rtk::Scene sc = old;
    sc.n_nodes = sc.n_tris = sc.n_verts = 0; sc.nrm = nullptr;
    if ((rc = install_scene(ctx, sc, &m)) != RT_OK) return rc;
    if (!old_nrm.empty()) {                                                        // smooth normals travel with their triangles
        std::vector<int> old_visit_of(nt, -1);
        for (size_t t = 0; t < old_perm.size(); ++t) old_visit_of[old_perm[t]] = (int)t;
        std::vector<float4> nn(ctx->tri_perm.size() * 3);
        for (size_t t = 0; t < ctx->tri_perm.size(); ++t) {
            const int ov = old_visit_of[order[ctx->tri_perm[t]]];
            for (int k = 0; k < 3; ++k) nn[3 * t + k] = ov >= 0 ? old_nrm[3 * (size_t)ov + k] : make_float4(0, 0, 0, 0);
        }
        if ((rc = upload(ctx, ctx->nrm, nn.data(), nn.size() * sizeof(float4))) != RT_OK) return rc;
        ctx->scene.nrm = static_cast<const float4 *>(ctx->nrm.p);
    }
    if (bvh_arr10_out) memcpy(bvh_arr10_out, arr.data(), arr.size() * sizeof(float));
    if (tri_order_out) memcpy(tri_order_out, order.data(), order.size() * sizeof(int));
    if (n_nodes_out) *n_nodes_out = n_nodes;
    ctx->build.install_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_install).count();
    return RT_OK;
}

int rt_mesh_rebuild(rt_ctx *ctx, float *bvh_arr10_out, int32_t *tri_order_out, int32_t *n_nodes_out) {
    return rt_mesh_rebuild_mode(ctx, RT_BVH_REFERENCE, bvh_arr10_out, tri_order_out, n_nodes_out);
}

int rt_mesh_build_stats(const rt_ctx *ctx, rt_build_stats *out) {
    if (!ctx || !out) return fail(nullptr, RT_ERR_INVALID, "bad arguments");
    *out = ctx->build;
    return RT_OK;
}
