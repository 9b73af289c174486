# SimSpreadHIP.jl -- the binding a SimSpread.jl maintainer would add to route the hot path
# (featurize -> construct -> spread -> predict -> clean!) through libsimspread_hip.so on an MI355X.
#
# NOT exercised in this repository's CI: Julia is not installed in the build container nor on the
# GPU box.  It is written against include/simspread_hip.h, the same ABI the Python mirror
# (simspread.jl_amd/) and the parity tests drive through ctypes.  No CUDA.jl, no dual backend.
#
# Usage inside SimSpread.jl (see INTEGRATION.md):
#     include("SimSpreadHIP.jl"); using .SimSpreadHIP
#     SimSpreadHIP.init(0)                                   # one process per GPU
#     g    = SimSpreadHIP.graph(Xtest.array, Xtrain.array, ytrain.array)      # construct (blocks only)
#     yhat = SimSpreadHIP.predict(g, :query; clean=false)    # Matrix{Float64}, Nq x Nt, column-major
module SimSpreadHIP

using SparseArrays

const LIB = get(ENV, "SIMSPREAD_HIP_LIB", joinpath(@__DIR__, "..", "simspread.jl_amd", "libsimspread_hip.so"))

const SS_MEM_HOST = Cint(0)
const SS_ROWS_QUERY = Cint(0)
const SS_ROWS_SOURCE = Cint(1)
const SS_LAYOUT_COLMAJOR = Cint(1)

struct SimSpreadHIPError <: Exception
    code::Cint
    msg::String
end

function check(rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:ss_last_error, LIB), Cstring, ()))
    # -1 (SS_EINVAL) is what the reference's @assert would have caught on the Julia side
    rc == -1 ? throw(AssertionError(msg)) : throw(SimSpreadHIPError(rc, msg))
end

init(device::Integer=0) = check(ccall((:ss_init, LIB), Cint, (Cint,), device))
shutdown() = check(ccall((:ss_shutdown, LIB), Cint, ()))

mutable struct Graph{T<:Union{Float32,Float64}}
    handle::Ptr{Cvoid}
    nq::Int
    ns::Int
    nf::Int
    nt::Int
    function Graph{T}(h, nq, ns, nf, nt) where {T}
        g = new{T}(h, nq, ns, nf, nt)
        finalizer(x -> ccall((:ss_graph_destroy, LIB), Cint, (Ptr{Cvoid},), x.handle), g)
        return g
    end
end

_sym(::Type{Float32}, name) = Symbol(name, "_f32")
_sym(::Type{Float64}, name) = Symbol(name, "_f64")

"""
    graph(Xq, Xs, Y; alpha=nothing, weighted=true, T=Float64)

Device-resident replacement of `construct` (src/core.jl:148-201,217-276,308-337): dense column-major
blocks `Xq = A[queries, features]`, `Xs = A[sources, features]`, `Y = A[sources, targets]`.
With `alpha` the featurize cutoff (src/core.jl:106-112) is fused into the on-device CSR assembly.
`Xq === nothing` builds the 3-layer graph of `construct(y, X)`.
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
    f = _sym(T, "ss_graph_create_dense")
    rc = if T === Float32
        ccall((:ss_graph_create_dense_f32, LIB), Cint,
              (Int64, Int64, Int64, Int64, Ptr{Float32}, Int64, Ptr{Float32}, Int64, Ptr{Float32}, Int64,
               Cint, Float32, Cint, Cint, Ref{Ptr{Cvoid}}),
              nq, ns, nf, nt, q, max(nq, 1), s, max(ns, 1), y, max(ns, 1),
              alpha === nothing ? 0 : 1, Float32(alpha === nothing ? 0 : alpha), weighted ? 1 : 0, SS_MEM_HOST, h)
    else
        ccall((:ss_graph_create_dense_f64, LIB), Cint,
              (Int64, Int64, Int64, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Int64,
               Cint, Float64, Cint, Cint, Ref{Ptr{Cvoid}}),
              nq, ns, nf, nt, q, max(nq, 1), s, max(ns, 1), y, max(ns, 1),
              alpha === nothing ? 0 : 1, Float64(alpha === nothing ? 0 : alpha), weighted ? 1 : 0, SS_MEM_HOST, h)
    end
    check(rc)
    return Graph{T}(h[], nq, ns, nf, nt)
end

"""
    graph(Xq::SparseMatrixCSC, Xs::SparseMatrixCSC, Y::SparseMatrixCSC; T=Float64)

Sparse inputs.  A `SparseMatrixCSC` is the 1-based CSR of its transpose, so the transposes are
materialised once (`sparse(X')`) and handed over with `index_base = 1`.
"""
function graph(Xq::SparseMatrixCSC, Xs::SparseMatrixCSC, Y::SparseMatrixCSC; T::Type=Float64)
    csr(M) = (t = sparse(M'); (Vector{Int64}(t.colptr), Vector{Int32}(t.rowval), Vector{T}(t.nzval)))
    (qp, qi, qv), (sp, si, sv), (yp, yi, yv) = csr(Xq), csr(Xs), csr(Y)
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
    predict(g, rows=:query; clean=false, range=nothing) -> Matrix{Float64}

The `rows x targets` block of `A * spread(B)^2` (src/core.jl:402-425,446-466), column-major like every
Julia matrix; `clean=true` fuses `clean!` (src/core.jl:478-484).  Results are widened to Float64 as the
reference does for `GPU=true` (src/core.jl:413).
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

k-fold cross-validation in one call (`fold_of_source[i] ∈ 1:k`, e.g. from `split`): row `i` is what the fold loop
`construct(y, X, members)` + `predict` (+ `clean!`) gives for source `i` when its fold is held out.
"""
function predict_kfold(g::Graph{T}, fold_of_source::AbstractVector{<:Integer}; clean::Bool=true) where {T}
    length(fold_of_source) == g.ns || throw(AssertionError("one fold index per source is needed"))
    folds = Vector{Int32}(fold_of_source .- 1)
    k = Int(maximum(folds)) + 1
    out = Matrix{T}(undef, g.ns, g.nt)
    rc = if T === Float32
        ccall((:ss_predict_kfold_f32, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Cint, Cint, Ptr{Float32}, Int64, Cint, Cint),
              g.handle, folds, k, clean ? 1 : 0, out, max(g.ns, 1), SS_LAYOUT_COLMAJOR, SS_MEM_HOST)
    else
        ccall((:ss_predict_kfold_f64, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Cint, Cint, Ptr{Float64}, Int64, Cint, Cint),
              g.handle, folds, k, clean ? 1 : 0, out, max(g.ns, 1), SS_LAYOUT_COLMAJOR, SS_MEM_HOST)
    end
    check(rc)
    return Matrix{Float64}(out)
end

"cutoff(X, alpha, weighted) on the device (src/core.jl:55-60)."
function cutoff(X::Matrix{Float64}, alpha::Float64, weighted::Bool=false)
    out = similar(X)
    check(ccall((:ss_cutoff_f64, LIB), Cint,
                (Ptr{Float64}, Int64, Int64, Int64, Float64, Cint, Ptr{Float64}, Int64, Cint),
                X, size(X, 1), size(X, 2), max(size(X, 1), 1), alpha, weighted ? 1 : 0, out, max(size(X, 1), 1), SS_MEM_HOST))
    return out
end

"""
    rank_metrics(y, yhat; alpha=20.0) -> (AuROC, AuPRC, BEDROC, validity_ratio)

The threshold-free metrics of `src/performance.jl:22-89,558-560` for one label / score vector, computed on the
device (sort + prefix sums); `SimSpread.AuROC(y, yhat)` etc. become one-liners over this call.
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

"""
    jaccard_similarity(X) -> Matrix

`1 .- pairwise(Jaccard(), X, dims=1)` (the similarity producer of the tutorial) on the device.
"""
function jaccard_similarity(X::Matrix{Float64})
    n, d = size(X)
    S = Matrix{Float64}(undef, n, n)
    check(ccall((:ss_similarity_jaccard_f64, LIB), Cint, (Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Int64, Cint),
                X, n, d, max(n, 1), S, max(n, 1), SS_MEM_HOST))
    return S
end

end # module
