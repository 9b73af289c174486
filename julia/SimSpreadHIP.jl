# SimSpreadHIP.jl -- raw `ccall` layer over libsimspread_hip.so (C ABI: include/simspread_hip.h), MI355X / gfx950.
#
# One Julia function per exported `ss_*` symbol, nothing else: argument conversion, the ccall, the error check.  The
# reference-compatible method table (featurize / construct / predict / clean! ... of src/SimSpread.jl:21-56) lives in
# SimSpreadDevice.jl, which is written on top of this module.
#
# NOT executed in this repository: Julia is installed neither in the build container nor on the GPU box.  What keeps the
# file honest is tests/test_julia_binding.py, which parses every ccall below and compares its name, return type and
# argument tuple with the C header and with the ctypes table of the Python mirror (simspread.jl_amd/_lib.py) that the
# parity tests drive.  No CUDA.jl, no dual backend.
module SimSpreadHIP

using SparseArrays

const LIB = get(ENV, "SIMSPREAD_HIP_LIB", joinpath(@__DIR__, "..", "simspread.jl_amd", "libsimspread_hip.so"))

const SS_MEM_HOST = Cint(0)
const SS_MEM_DEVICE = Cint(1)
const SS_ROWS_QUERY = Cint(0)
const SS_ROWS_SOURCE = Cint(1)
const SS_LAYOUT_ROWMAJOR = Cint(0)
const SS_LAYOUT_COLMAJOR = Cint(1)

struct SimSpreadHIPError <: Exception
    code::Cint
    msg::String
end

# ------------------------------------------------------------------------------------------------ runtime
version() = ccall((:ss_version, LIB), Cint, ())
last_error() = unsafe_string(ccall((:ss_last_error, LIB), Cstring, ()))
source_hash() = unsafe_string(ccall((:ss_source_hash, LIB), Cstring, ()))
device_count() = ccall((:ss_device_count, LIB), Cint, ())

function check(rc::Cint)
    rc == 0 && return nothing
    msg = last_error()
    # -1 (SS_EINVAL) is what the reference's @assert would have caught on the Julia side
    rc == -1 ? throw(AssertionError(msg)) : throw(SimSpreadHIPError(rc, msg))
end

"Bind this process to one GPU (replaces the reference's `GPU::Bool` switch, src/core.jl:402,404,446,448)."
init(device::Integer=0) = check(ccall((:ss_init, LIB), Cint, (Cint,), device))
shutdown() = check(ccall((:ss_shutdown, LIB), Cint, ()))
set_stream(stream::Ptr{Cvoid}) = check(ccall((:ss_set_stream, LIB), Cint, (Ptr{Cvoid},), stream))
reset_stream() = check(ccall((:ss_reset_stream, LIB), Cint, ()))
synchronize() = check(ccall((:ss_synchronize, LIB), Cint, ()))
timing_hold(enable::Bool) = check(ccall((:ss_timing_hold, LIB), Cint, (Cint,), enable ? 1 : 0))

"Stage timings (ms) of the last predict / spmm call of this task's thread: total, transfer, spmm, epilogue, h2d, d2h, #spmm, #transfer."
function timing_last()
    ms = zeros(Float64, 8)
    check(ccall((:ss_timing_last, LIB), Cint, (Ptr{Float64}, Cint), ms, 8))
    return (total_ms=ms[1], transfer_ms=ms[2], spmm_ms=ms[3], epilogue_ms=ms[4], h2d_ms=ms[5], d2h_ms=ms[6],
            spmm_launches=Int(ms[7]), transfer_launches=Int(ms[8]))
end

"Kernel tags the last predict / spmm call went through (e.g. \"transfer_dense_bf16_ring\", \"spmm_sell\")."
function path_last()
    buf = zeros(UInt8, 512)
    check(ccall((:ss_path_last, LIB), Cint, (Ptr{UInt8}, Cint), buf, 512))
    return split(unsafe_string(pointer(buf)), ","; keepempty=false)
end

# ------------------------------------------------------------------------------------------------ final score gather (RCCL)
"128 bytes created by rank 0; hand them to the other processes (MPI.bcast, Distributed.jl, ...)."
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:ss_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id))
    return id
end
comm_init(id::Vector{UInt8}, rank::Integer, nranks::Integer) =
    check(ccall((:ss_comm_init, LIB), Cint, (Ptr{UInt8}, Cint, Cint), id, rank, nranks))
comm_destroy() = check(ccall((:ss_comm_destroy, LIB), Cint, ()))
function comm_info()
    rank, nranks = Ref{Cint}(0), Ref{Cint}(0)
    check(ccall((:ss_comm_info, LIB), Cint, (Ref{Cint}, Ref{Cint}), rank, nranks))
    return Int(rank[]), Int(nranks[])
end

"""
    gather_rows(local::Ptr, ncols, counts, full::Ptr; root=-1, T=Float32)

Direct exchange of finished score rows between the ranks (device pointers, row-major blocks): rank r contributes
`counts[r+1]` rows; `root = -1`: every rank receives the full matrix.
"""
function gather_rows(local_::Ptr{Cvoid}, ncols::Integer, counts::Vector{Int64}, full::Ptr{Cvoid}; root::Integer=-1, T::Type=Float32)
    rc = if T === Float32
        ccall((:ss_gather_rows_f32, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Cvoid}, Cint), local_, ncols, counts, full, root)
    else
        ccall((:ss_gather_rows_f64, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Cvoid}, Cint), local_, ncols, counts, full, root)
    end
    check(rc)
end

# ------------------------------------------------------------------------------------------------ cutoff / k / spread
"cutoff(X, alpha, weighted) on the device (src/core.jl:37-43,55-60): x >= alpha ? (weighted ? x : 1) : 0."
function cutoff(X::Matrix{Float64}, alpha::Float64, weighted::Bool=false)
    out = similar(X)
    ld = max(size(X, 1), 1)
    check(ccall((:ss_cutoff_f64, LIB), Cint,
                (Ptr{Float64}, Int64, Int64, Int64, Float64, Cint, Ptr{Float64}, Int64, Cint),
                X, size(X, 1), size(X, 2), ld, alpha, weighted ? 1 : 0, out, ld, SS_MEM_HOST))
    return out
end
function cutoff(X::Matrix{Float32}, alpha::Float32, weighted::Bool=false)
    out = similar(X)
    ld = max(size(X, 1), 1)
    check(ccall((:ss_cutoff_f32, LIB), Cint,
                (Ptr{Float32}, Int64, Int64, Int64, Float32, Cint, Ptr{Float32}, Int64, Cint),
                X, size(X, 1), size(X, 2), ld, alpha, weighted ? 1 : 0, out, ld, SS_MEM_HOST))
    return out
end

"k(G): number of non-zeros of every row (src/graphs.jl:9-11) -> Vector{Int64}."
function row_degree(G::Matrix{Float64})
    deg = Vector{Int64}(undef, size(G, 1))
    check(ccall((:ss_row_degree_f64, LIB), Cint, (Ptr{Float64}, Int64, Int64, Int64, Ptr{Int64}, Cint),
                G, size(G, 1), size(G, 2), max(size(G, 1), 1), deg, SS_MEM_HOST))
    return deg
end
function row_degree(G::Matrix{Float32})
    deg = Vector{Int64}(undef, size(G, 1))
    check(ccall((:ss_row_degree_f32, LIB), Cint, (Ptr{Float32}, Int64, Int64, Int64, Ptr{Int64}, Cint),
                G, size(G, 1), size(G, 2), max(size(G, 1), 1), deg, SS_MEM_HOST))
    return deg
end

"spread(G): W[i,j] = G[i,j] / k(i), rows of degree 0 give 0 (src/core.jl:365-371)."
function spread(G::Matrix{Float64})
    W = similar(G)
    ld = max(size(G, 1), 1)
    check(ccall((:ss_spread_f64, LIB), Cint, (Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Int64, Cint),
                G, size(G, 1), size(G, 2), ld, W, ld, SS_MEM_HOST))
    return W
end
function spread(G::Matrix{Float32})
    W = similar(G)
    ld = max(size(G, 1), 1)
    check(ccall((:ss_spread_f32, LIB), Cint, (Ptr{Float32}, Int64, Int64, Int64, Ptr{Float32}, Int64, Cint),
                G, size(G, 1), size(G, 2), ld, W, ld, SS_MEM_HOST))
    return W
end

"`1 .- pairwise(Jaccard(), X, dims=1)` (docs/src/tutorial/fishers-flowers.jl:66) on the device."
function jaccard_similarity(X::Matrix{Float64})
    n, d = size(X)
    S = Matrix{Float64}(undef, n, n)
    check(ccall((:ss_similarity_jaccard_f64, LIB), Cint, (Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Int64, Cint),
                X, n, d, max(n, 1), S, max(n, 1), SS_MEM_HOST))
    return S
end
function jaccard_similarity(X::Matrix{Float32})
    n, d = size(X)
    S = Matrix{Float32}(undef, n, n)
    check(ccall((:ss_similarity_jaccard_f32, LIB), Cint, (Ptr{Float32}, Int64, Int64, Int64, Ptr{Float32}, Int64, Cint),
                X, n, d, max(n, 1), S, max(n, 1), SS_MEM_HOST))
    return S
end

# ------------------------------------------------------------------------------------------------ graph handles
mutable struct Graph{T<:Union{Float32,Float64}}
    handle::Ptr{Cvoid}
    nq::Int
    ns::Int
    nf::Int
    nt::Int
    function Graph{T}(h, nq, ns, nf, nt) where {T}
        g = new{T}(h, nq, ns, nf, nt)
        finalizer(destroy!, g)
        return g
    end
end

function destroy!(g::Graph)
    if g.handle != C_NULL
        ccall((:ss_graph_destroy, LIB), Cint, (Ptr{Cvoid},), g.handle)
        g.handle = C_NULL
    end
    return nothing
end

# A SparseMatrixCSC is the 1-based CSR of its transpose: hand over the transposes with index_base = 1.
_csr(M::SparseMatrixCSC, ::Type{T}) where {T} = (t = sparse(M'); (Vector{Int64}(t.colptr), Vector{Int32}(t.rowval), Vector{T}(t.nzval)))

"""
    graph(Xq, Xs, Y; alpha=nothing, weighted=true, T=Float64)

Device-resident replacement of `construct` (src/core.jl:148-201,217-276,308-337): dense column-major blocks
`Xq = A[queries, features]`, `Xs = A[sources, features]`, `Y = A[sources, targets]`.  With `alpha` the featurize
cutoff (src/core.jl:106-112) is fused into the on-device CSR assembly.  `Xq === nothing`: the 3-layer graph of
`construct(y, X)`.
"""
function graph(Xq::Union{Nothing,AbstractMatrix}, Xs::AbstractMatrix, Y::AbstractMatrix;
               alpha=nothing, weighted::Bool=true, T::Type=Float64)
    size(Y, 1) == size(Xs, 1) || throw(AssertionError("Labels and features have different number of source nodes"))
    Xq === nothing || size(Xq, 2) == size(Xs, 2) ||
        throw(AssertionError("Number of features between test and training sets doesn't match"))
    q = Xq === nothing ? Matrix{T}(undef, 0, size(Xs, 2)) : Matrix{T}(Xq)
    s, y = Matrix{T}(Xs), Matrix{T}(Y)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    nq, ns, nf, nt = size(q, 1), size(s, 1), size(s, 2), size(y, 2)
    cut, a = alpha === nothing ? 0 : 1, alpha === nothing ? zero(T) : T(alpha)
    rc = if T === Float32
        ccall((:ss_graph_create_dense_f32, LIB), Cint,
              (Int64, Int64, Int64, Int64, Ptr{Float32}, Int64, Ptr{Float32}, Int64, Ptr{Float32}, Int64,
               Cint, Float32, Cint, Cint, Ref{Ptr{Cvoid}}),
              nq, ns, nf, nt, q, max(nq, 1), s, max(ns, 1), y, max(ns, 1), cut, a, weighted ? 1 : 0, SS_MEM_HOST, h)
    else
        ccall((:ss_graph_create_dense_f64, LIB), Cint,
              (Int64, Int64, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64,
               Cint, Float64, Cint, Cint, Ref{Ptr{Cvoid}}),
              nq, ns, nf, nt, q, max(nq, 1), s, max(ns, 1), y, max(ns, 1), cut, a, weighted ? 1 : 0, SS_MEM_HOST, h)
    end
    check(rc)
    return Graph{T}(h[], nq, ns, nf, nt)
end

"Sparse blocks (`SparseMatrixCSC`)."
function graph(Xq::SparseMatrixCSC, Xs::SparseMatrixCSC, Y::SparseMatrixCSC; T::Type=Float64)
    (qp, qi, qv), (sp, si, sv), (yp, yi, yv) = _csr(Xq, T), _csr(Xs, T), _csr(Y, T)
    nq, ns, nf, nt = size(Xq, 1), size(Xs, 1), size(Xs, 2), size(Y, 2)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = if T === Float32
        ccall((:ss_graph_create_csr_f32, LIB), Cint,
              (Int64, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Float32}, Ptr{Int64}, Ptr{Int32}, Ptr{Float32},
               Ptr{Int64}, Ptr{Int32}, Ptr{Float32}, Cint, Cint, Ref{Ptr{Cvoid}}),
              nq, ns, nf, nt, qp, qi, qv, sp, si, sv, yp, yi, yv, 1, SS_MEM_HOST, h)
    else
        ccall((:ss_graph_create_csr_f64, LIB), Cint,
              (Int64, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Float64}, Ptr{Int64}, Ptr{Int32}, Ptr{Float64},
               Ptr{Int64}, Ptr{Int32}, Ptr{Float64}, Cint, Cint, Ref{Ptr{Cvoid}}),
              nq, ns, nf, nt, qp, qi, qv, sp, si, sv, yp, yi, yv, 1, SS_MEM_HOST, h)
    end
    check(rc)
    return Graph{T}(h[], nq, ns, nf, nt)
end

"""
    graph_similarity(Sq, Ss, Y::SparseMatrixCSC; alpha, weighted=true)

Dense-similarity regime (thresholded similarity too full for CSR): raw similarities `Sq` (nq x ns, or `nothing`) and
`Ss` (ns x ns) stay dense on the device, featurize's cutoff is applied inside the matrix-core product
(`T=Float32`: bf16 planes, `T=Float64`: the fp64 matrix instruction).
"""
function graph_similarity(Sq::Union{Nothing,AbstractMatrix}, Ss::AbstractMatrix, Y::SparseMatrixCSC;
                          alpha::Real, weighted::Bool=true, T::Type=Float32)
    ns = size(Ss, 1)
    q = Sq === nothing ? Matrix{T}(undef, 0, ns) : Matrix{T}(Sq)
    s = Matrix{T}(Ss)
    yp, yi, yv = _csr(Y, T)
    nq, nt = size(q, 1), size(Y, 2)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = if T === Float32
        ccall((:ss_graph_create_similarity_f32, LIB), Cint,
              (Int64, Int64, Int64, Ptr{Float32}, Int64, Ptr{Float32}, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Float32},
               Cint, Float32, Cint, Cint, Ref{Ptr{Cvoid}}),
              nq, ns, nt, q, max(nq, 1), s, max(ns, 1), yp, yi, yv, 1, Float32(alpha), weighted ? 1 : 0, SS_MEM_HOST, h)
    else
        ccall((:ss_graph_create_similarity_f64, LIB), Cint,
              (Int64, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Float64},
               Cint, Float64, Cint, Cint, Ref{Ptr{Cvoid}}),
              nq, ns, nt, q, max(nq, 1), s, max(ns, 1), yp, yi, yv, 1, Float64(alpha), weighted ? 1 : 0, SS_MEM_HOST, h)
    end
    check(rc)
    return Graph{T}(h[], nq, ns, ns, nt)
end

"""
    graph_general(L, B, Wt; T=Float64)

Any caller-built adjacency pair (src/core.jl:402-425 accepts arbitrary named `A`, `B`): `L = A[rows of y, :]` (nr x n),
`B` (n x n), `Wt = B[:, cols of y]'` (nc x n), all `SparseMatrixCSC`.  `predict(g, :query)` then returns the
nr x nc block of `A * spread(B)^2`.
"""
function graph_general(L::SparseMatrixCSC, B::SparseMatrixCSC, Wt::SparseMatrixCSC; T::Type=Float64)
    n, nr, nc = size(B, 1), size(L, 1), size(Wt, 1)
    (lp, li, lv), (bp, bi, bv), (wp, wi, wv) = _csr(L, T), _csr(B, T), _csr(Wt, T)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = if T === Float32
        ccall((:ss_graph_create_general_f32, LIB), Cint,
              (Int64, Int64, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Float32}, Ptr{Int64}, Ptr{Int32}, Ptr{Float32},
               Ptr{Int64}, Ptr{Int32}, Ptr{Float32}, Cint, Cint, Ref{Ptr{Cvoid}}),
              n, nr, nc, lp, li, lv, bp, bi, bv, wp, wi, wv, 1, SS_MEM_HOST, h)
    else
        ccall((:ss_graph_create_general_f64, LIB), Cint,
              (Int64, Int64, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Float64}, Ptr{Int64}, Ptr{Int32}, Ptr{Float64},
               Ptr{Int64}, Ptr{Int32}, Ptr{Float64}, Cint, Cint, Ref{Ptr{Cvoid}}),
              n, nr, nc, lp, li, lv, bp, bi, bv, wp, wi, wv, 1, SS_MEM_HOST, h)
    end
    check(rc)
    return Graph{T}(h[], nr, n, n, nc)
end

"sizes after dropping stored zeros: (nq, ns, nf, nt, nnz(Xq), nnz(Xs), nnz(Ys))."
function graph_info(g::Graph)
    sizes = Vector{Int64}(undef, 7)
    check(ccall((:ss_graph_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), g.handle, sizes))
    return sizes
end

"Count degrees of the query-free graph B: (kf, ks, kt)."
function graph_degrees(g::Graph)
    kf, ks, kt = Vector{Int64}(undef, g.nf), Vector{Int64}(undef, g.ns), Vector{Int64}(undef, g.nt)
    check(ccall((:ss_graph_degrees, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), g.handle, kf, ks, kt))
    return kf, ks, kt
end

# ------------------------------------------------------------------------------------------------ predict
"""
    predict(g, rows=:query; clean=false, range=nothing) -> Matrix{Float64}

The `rows x targets` block of `A * spread(B)^2` (src/core.jl:402-425,446-466), column-major like every Julia matrix;
`clean=true` fuses `clean!` (src/core.jl:478-484).  Float32 results are widened to Float64 as the reference does for
`GPU=true` (src/core.jl:413).
"""
function predict(g::Graph{T}, rows::Symbol=:query; clean::Bool=false, range=nothing) where {T}
    kind = rows === :query ? SS_ROWS_QUERY : SS_ROWS_SOURCE
    n = rows === :query ? g.nq : g.ns
    lo, hi = range === nothing ? (0, n) : (first(range) - 1, last(range))
    out = Matrix{T}(undef, hi - lo, g.nt)
    rc = if T === Float32
        ccall((:ss_predict_f32, LIB), Cint, (Ptr{Cvoid}, Cint, Int64, Int64, Cint, Ptr{Float32}, Int64, Cint, Cint),
              g.handle, kind, lo, hi, clean ? 1 : 0, out, max(hi - lo, 1), SS_LAYOUT_COLMAJOR, SS_MEM_HOST)
    else
        ccall((:ss_predict_f64, LIB), Cint, (Ptr{Cvoid}, Cint, Int64, Int64, Cint, Ptr{Float64}, Int64, Cint, Cint),
              g.handle, kind, lo, hi, clean ? 1 : 0, out, max(hi - lo, 1), SS_LAYOUT_COLMAJOR, SS_MEM_HOST)
    end
    check(rc)
    return Matrix{Float64}(out)
end

"""
    predict_loo(g; clean=true, range=nothing)

All leave-one-out folds of `construct(y, X, [source_i])` + `predict` (+ `clean!`) from one resident graph
(`g = graph(nothing, X, y)` with square `X`), rows `range` (default all sources).
"""
function predict_loo(g::Graph{T}; clean::Bool=true, range=nothing) where {T}
    lo, hi = range === nothing ? (0, g.ns) : (first(range) - 1, last(range))
    out = Matrix{T}(undef, hi - lo, g.nt)
    rc = if T === Float32
        ccall((:ss_predict_loo_f32, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Cint, Ptr{Float32}, Int64, Cint, Cint),
              g.handle, lo, hi, clean ? 1 : 0, out, max(hi - lo, 1), SS_LAYOUT_COLMAJOR, SS_MEM_HOST)
    else
        ccall((:ss_predict_loo_f64, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Cint, Ptr{Float64}, Int64, Cint, Cint),
              g.handle, lo, hi, clean ? 1 : 0, out, max(hi - lo, 1), SS_LAYOUT_COLMAJOR, SS_MEM_HOST)
    end
    check(rc)
    return Matrix{Float64}(out)
end

"""
    predict_kfold(g, fold_of_source; clean=true)

k-fold cross-validation in one call (`fold_of_source[i] in 1:k`, e.g. from `split`): row `i` is what the fold loop
`construct(y, X, members)` + `predict` (+ `clean!`) gives for source `i` when its fold is held out.
"""
function predict_kfold(g::Graph{T}, fold_of_source::AbstractVector{<:Integer}; clean::Bool=true) where {T}
    length(fold_of_source) == g.ns || throw(AssertionError("one fold index per source is needed"))
    folds = Vector{Int32}(fold_of_source .- 1)
    nfolds = Int(maximum(folds)) + 1
    out = Matrix{T}(undef, g.ns, g.nt)
    rc = if T === Float32
        ccall((:ss_predict_kfold_f32, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Cint, Cint, Ptr{Float32}, Int64, Cint, Cint),
              g.handle, folds, nfolds, clean ? 1 : 0, out, max(g.ns, 1), SS_LAYOUT_COLMAJOR, SS_MEM_HOST)
    else
        ccall((:ss_predict_kfold_f64, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Cint, Cint, Ptr{Float64}, Int64, Cint, Cint),
              g.handle, folds, nfolds, clean ? 1 : 0, out, max(g.ns, 1), SS_LAYOUT_COLMAJOR, SS_MEM_HOST)
    end
    check(rc)
    return Matrix{Float64}(out)
end

# ------------------------------------------------------------------------------------------------ ranked evaluation
"""
    topl(scores::Matrix{Float32}, L) -> (idx, val)

The L best targets of every row of a `rows x targets` score matrix in `sortperm(yhat, rev=true)` order
(src/performance.jl:308-385: recallatL / precisionatL need nothing else).  The library wants the block row-major,
i.e. the transpose of the Julia matrix as it lies in memory.  `idx` is 1-based.
"""
function topl(scores::Matrix{Float32}, L::Integer)
    nrows, ncols = size(scores)
    rowmajor = Matrix{Float32}(scores')          # ncols x nrows column-major == nrows x ncols row-major
    idx = Matrix{Int32}(undef, L, nrows)
    val = Matrix{Float32}(undef, L, nrows)
    check(ccall((:ss_topl_f32, LIB), Cint, (Ptr{Float32}, Int64, Int64, Int64, Cint, Ptr{Int32}, Ptr{Float32}, Cint),
                rowmajor, nrows, ncols, ncols, L, idx, val, SS_MEM_HOST))
    return Matrix{Int}(idx') .+ 1, Matrix{Float32}(val')
end

"""
    rank_metrics(y, yhat; alpha=20.0) -> (AuROC, AuPRC, BEDROC, validity_ratio)

The threshold-free metrics of src/performance.jl:22-89,558-560 for one label / score vector, on the device.
"""
function rank_metrics(y::AbstractVector, yhat::AbstractVector; alpha::Float64=20.0)
    length(y) == length(yhat) || throw(AssertionError("The number of scores must be equal to the number of labels"))
    labels = Vector{UInt8}(y .!= 0)
    scores = Vector{Float32}(yhat)
    out = Vector{Float64}(undef, 4)
    check(ccall((:ss_rank_metrics_f32, LIB), Cint, (Ptr{UInt8}, Ptr{Float32}, Int64, Float64, Ptr{Float64}, Cint),
                labels, scores, length(scores), alpha, out, SS_MEM_HOST))
    return (AuROC=out[1], AuPRC=out[2], BEDROC=out[3], validity_ratio=out[4])
end

# ------------------------------------------------------------------------------------------------ raw W*R SpMM
mutable struct SpMat{T<:Union{Float32,Float64}}
    handle::Ptr{Cvoid}
    rows::Int
    cols::Int
    function SpMat{T}(h, rows, cols) where {T}
        w = new{T}(h, rows, cols)
        finalizer(destroy!, w)
        return w
    end
end

function destroy!(w::SpMat)
    if w.handle != C_NULL
        ccall((:ss_spmat_destroy, LIB), Cint, (Ptr{Cvoid},), w.handle)
        w.handle = C_NULL
    end
    return nothing
end

"One sparse operand W of F = W*R, device resident (kernel unit tests and the roofline benchmark)."
function spmat(W::SparseMatrixCSC; T::Type=Float64)
    p, i, v = _csr(W, T)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = if T === Float32
        ccall((:ss_spmat_create_csr_f32, LIB), Cint,
              (Int64, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Float32}, Cint, Cint, Ref{Ptr{Cvoid}}),
              size(W, 1), size(W, 2), p, i, v, 1, SS_MEM_HOST, h)
    else
        ccall((:ss_spmat_create_csr_f64, LIB), Cint,
              (Int64, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Float64}, Cint, Cint, Ref{Ptr{Cvoid}}),
              size(W, 1), size(W, 2), p, i, v, 1, SS_MEM_HOST, h)
    end
    check(rc)
    return SpMat{T}(h[], size(W, 1), size(W, 2))
end

"F = W * R for a dense column-major R (cols(W) x B)."
function spmm(w::SpMat{T}, R::Matrix{T}) where {T}
    size(R, 1) == w.cols || throw(DimensionMismatch("R must have cols(W) rows"))
    B = size(R, 2)
    F = Matrix{T}(undef, w.rows, B)
    rc = if T === Float32
        ccall((:ss_spmm_f32, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Int64, Int64, Cint, Ptr{Float32}, Int64, Cint, Cint),
              w.handle, R, B, max(w.cols, 1), SS_LAYOUT_COLMAJOR, F, max(w.rows, 1), SS_LAYOUT_COLMAJOR, SS_MEM_HOST)
    else
        ccall((:ss_spmm_f64, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Cint, Ptr{Float64}, Int64, Cint, Cint),
              w.handle, R, B, max(w.cols, 1), SS_LAYOUT_COLMAJOR, F, max(w.rows, 1), SS_LAYOUT_COLMAJOR, SS_MEM_HOST)
    end
    check(rc)
    return F
end

"Algorithmic bytes and flops of one spmm call with B columns (SURVEY.md 8d)."
function spmat_cost(w::SpMat, B::Integer)
    bytes, flops = Ref{Float64}(0.0), Ref{Float64}(0.0)
    check(ccall((:ss_spmat_cost, LIB), Cint, (Ptr{Cvoid}, Int64, Ref{Float64}, Ref{Float64}), w.handle, B, bytes, flops))
    return bytes[], flops[]
end

end # module
