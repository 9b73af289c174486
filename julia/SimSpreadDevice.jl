# SimSpreadDevice.jl -- the reference's method table (src/SimSpread.jl:21-56) over libsimspread_hip.so.
#
#     k, cutoff, featurize, construct (x4), spread, predict (x3), clean!
#
# keep their names, argument meaning, return conventions and assertion messages (src/core.jl:37-43,55-60,106-112,
# 148-201,217-276,294-296,308-337,365-380,402-425,446-466,478-484; src/graphs.jl:9-11); the arithmetic runs in
# hand-written gfx950 kernels behind the C ABI (SimSpreadHIP.jl, include/simspread_hip.h).  A maintainer wires it in
# by replacing `include("core.jl")`'s hot-path methods with `include("SimSpreadDevice.jl"); using .SimSpreadDevice`
# (INTEGRATION.md).  No CUDA.jl.
#
# Differences a caller can see, all deliberate:
#   * `construct` returns `Network`s instead of dense named N x N matrices: the same node order
#     [queries; sources; features; targets] and names, but only the three non-zero blocks are stored (on the host and,
#     at first use, on the device).  `Matrix(net)` / `net.array` still materialise the reference's dense matrix;
#   * `predict(...; GPU=true)` computes in fp32 on the MI355X and widens to Float64 (what the reference's `GPU=true`
#     did through CUDA.jl, src/core.jl:404,413); `GPU=false` computes in fp64 -- also on the device.  There is no CPU
#     path in this module: the reference's own Julia code remains the CPU fallback;
#   * caller-built dense `NamedMatrix` pairs `(A, B)` (as in test/runtests.jl:120-158) are accepted too and go through
#     the general entry point.
#
# NOT executed in this repository (no Julia in the build container or on the GPU box); the Python mirror
# simspread.jl_amd/core.py implements the same table over the same ABI and is what tests/ drive.
module SimSpreadDevice

using NamedArrays
using SparseArrays

include("SimSpreadHIP.jl")
using .SimSpreadHIP

export k, cutoff, cutoff!, featurize, featurize!, construct, spread, predict, clean!, Network, predict_loo, predict_kfold

const NamedMatrix = NamedArrays.NamedMatrix

# ------------------------------------------------------------------------------------------------ k / cutoff / featurize / spread
"k(G): node degrees = number of non-zeros per row (src/graphs.jl:9-11); `N x 1` like `mapslices(k, G; dims=2)`."
k(vi::Integer, G::AbstractMatrix) = count(!iszero, G[vi, :])
k(ei::AbstractVector) = count(!iszero, ei)
k(G::AbstractMatrix) = reshape(SimSpreadHIP.row_degree(Matrix{Float64}(G)), :, 1)

"cutoff(x, alpha, weighted): `x >= alpha ? (weighted ? x : 1.0) : 0.0` (src/core.jl:37-43,55-60)."
function cutoff(x::T, alpha::T, weighted::Bool=false) where {T<:AbstractFloat}
    weight = weighted ? x : 1.0
    return x >= alpha ? weight : 0.0
end
function cutoff(X::AbstractVecOrMat{T}, alpha::T, weighted::Bool=false) where {T<:AbstractFloat}
    M = X isa AbstractVector ? reshape(Vector{Float64}(X), :, 1) : Matrix{Float64}(X)
    out = SimSpreadHIP.cutoff(M, Float64(alpha), weighted)
    return X isa AbstractVector ? vec(out) : out
end
# The reference's cutoff! rebinds a local / discards the broadcast result (src/core.jl:72-75,87-89), i.e. it leaves its
# argument untouched; SURVEY.md 3.2 quirk (8) says to match `cutoff` semantics instead of copying the no-op.
function cutoff!(X::AbstractVecOrMat{T}, alpha::T, weighted::Bool=false) where {T<:AbstractFloat}
    X .= cutoff(X, alpha, weighted)
    return X
end

"featurize(X, alpha, weighted): similarity cutoff + feature columns renamed `\"f\" * name` (src/core.jl:106-112)."
function featurize(X::NamedArray, alpha::AbstractFloat, weighted::Bool=true)
    Xp = copy(X)
    Xp.array = cutoff(Matrix{Float64}(X.array), Float64(alpha), weighted)
    setnames!(Xp, ["f$f" for f in names(Xp, 2)], 2)
    return Xp
end
function featurize!(X::NamedArray, alpha::AbstractFloat, weighted::Bool=true)   # src/core.jl:129-132
    X.array = cutoff(Matrix{Float64}(X.array), Float64(alpha), weighted)
    setnames!(X, ["f$f" for f in names(X, 2)], 2)
    return X
end

"spread(G): `W = G ./ k(G)` with zero-degree rows giving 0 (src/core.jl:365-371,373,375-380)."
spread(G::AbstractMatrix{Float64}) = SimSpreadHIP.spread(Matrix{Float64}(G))
spread(G::AbstractMatrix{Bool}) = spread(Matrix{Float64}(G))
function spread(G::NamedMatrix)
    W = copy(G)
    W.array = spread(Matrix{Float64}(G.array))
    return W
end

# ------------------------------------------------------------------------------------------------ Network
"""
What `construct` returns in place of the dense named N x N adjacency matrix: node names in the reference's order
[queries; sources; features; targets] (test/runtests.jl:97-98) and the three non-zero blocks
`Xq = A[queries, features]`, `Xs = A[sources, features]`, `Ys = A[sources, targets]` (src/core.jl:165-187).
`kind` is `:A` (full graph), `:B` (query rows/columns zeroed, src/core.jl:196-198) or `:single` (3-layer graph,
src/core.jl:308-337).  The A and B of one `construct` call share their device handles.
"""
struct Network
    kind::Symbol
    queries::Vector{String}
    sources::Vector{String}
    features::Vector{String}
    targets::Vector{String}
    Xq::Matrix{Float64}
    Xs::Matrix{Float64}
    Ys::Matrix{Float64}
    dev::Dict{DataType,SimSpreadHIP.Graph}
end

Base.names(N::Network) = (n = vcat(N.queries, N.sources, N.features, N.targets); [n, n])
Base.names(N::Network, d::Integer) = vcat(N.queries, N.sources, N.features, N.targets)
Base.size(N::Network) = (n = length(N.queries) + length(N.sources) + length(N.features) + length(N.targets); (n, n))
Base.size(N::Network, d::Integer) = size(N)[d]

"The reference's dense adjacency matrix of this network (for code that still indexes `A.array`)."
function Base.Matrix(N::Network)
    nq, ns, nf, nt = length(N.queries), length(N.sources), length(N.features), length(N.targets)
    n = nq + ns + nf + nt
    A = zeros(n, n)
    os, of, ot = nq, nq + ns, nq + ns + nf
    if N.kind != :B && nq > 0
        A[1:nq, of+1:of+nf] = N.Xq
        A[of+1:of+nf, 1:nq] = N.Xq'
    end
    A[os+1:os+ns, of+1:of+nf] = N.Xs
    A[os+1:os+ns, ot+1:ot+nt] = N.Ys
    A[of+1:of+nf, os+1:os+ns] = N.Xs'
    A[ot+1:ot+nt, os+1:os+ns] = N.Ys'
    return A
end
Base.getproperty(N::Network, s::Symbol) = s === :array ? Matrix(N) : getfield(N, s)
NamedArrays.NamedArray(N::Network) = NamedArray(Matrix(N), (names(N, 1), names(N, 2)))

function device(N::Network, ::Type{T}) where {T<:Union{Float32,Float64}}
    get!(N.dev, T) do
        SimSpreadHIP.graph(isempty(N.queries) ? nothing : N.Xq, N.Xs, N.Ys; T=T)
    end
end

# The reference compares the two sorted name vectors element-wise (src/core.jl:156,231,314): unequal lengths throw a
# DimensionMismatch from the broadcast (unless one side has one element) -- kept, it is part of the observable contract.
_names_differ(features, sources) = all(sort(features) .!= sort(sources))

# ------------------------------------------------------------------------------------------------ construct
"construct(y, X, queries): k-fold / leave-one-out form (src/core.jl:148-201)."
function construct(y::NamedMatrix, X::NamedMatrix, queries::AbstractVector)
    @assert size(y, 1) == size(X, 1) "Labels and features have different number of source nodes"
    features = [f for f in names(X, 2) if lstrip(f, 'f') ∉ queries]
    sources = [d for d in names(X, 1) if d ∉ queries]
    targets = names(y, 2)
    @assert _names_differ(features, sources) "Source and Features nodes have the same names!"
    Xq = Matrix{Float64}(X[queries, features].array)
    Xs = Matrix{Float64}(X[sources, features].array)
    Ys = Matrix{Float64}(y[sources, targets].array)
    dev = Dict{DataType,SimSpreadHIP.Graph}()
    q, s, f, t = string.(queries), string.(sources), string.(features), string.(targets)
    return Network(:A, q, s, f, t, Xq, Xs, Ys, dev), Network(:B, q, s, f, t, Xq, Xs, Ys, dev)
end

"construct((ytrain, ytest), (Xtrain, Xtest)): time-split form (src/core.jl:217-276)."
function construct(ys::T, Xs::T) where {T<:Tuple{NamedMatrix,NamedMatrix}}
    ytrain, ytest = ys
    Xtrain, Xtest = Xs
    @assert size(ytrain, 2) == size(ytest, 2) "Number of targets between test and training sets doesn't match"
    @assert size(Xtrain, 2) == size(Xtest, 2) "Number of features between test and training sets doesn't match"
    features = names(Xtrain, 2)
    sources = names(ytrain, 1)
    targets = names(ytrain, 2)
    queries = names(ytest, 1)
    @assert _names_differ(features, sources) "Features and drugs have the same names!"
    Xq, Xsrc, Ysrc = Matrix{Float64}(Xtest.array), Matrix{Float64}(Xtrain.array), Matrix{Float64}(ytrain.array)
    dev = Dict{DataType,SimSpreadHIP.Graph}()
    q, s, f, t = string.(queries), string.(sources), string.(features), string.(targets)
    return Network(:A, q, s, f, t, Xq, Xsrc, Ysrc, dev), Network(:B, q, s, f, t, Xq, Xsrc, Ysrc, dev)
end

"construct(ytrain, ytest, Xtrain, Xtest) (src/core.jl:294-296)."
construct(ytrain::T, ytest::T, Xtrain::T, Xtest::T) where {T<:NamedMatrix} = construct((ytrain, ytest), (Xtrain, Xtest))

"construct(y, X): 3-layer source-feature-target graph (src/core.jl:308-337)."
function construct(y::NamedMatrix, X::NamedMatrix)
    features = names(X, 2)
    sources = names(y, 1)
    targets = names(y, 2)
    @assert _names_differ(features, sources) "Source and feature nodes have the same names"
    Xs, Ys = Matrix{Float64}(X.array), Matrix{Float64}(y.array)
    return Network(:single, String[], string.(sources), string.(features), string.(targets),
                   zeros(0, length(features)), Xs, Ys, Dict{DataType,SimSpreadHIP.Graph}())
end

# ------------------------------------------------------------------------------------------------ predict
# rows/columns of `y` looked up in the network: query rows come from ss_predict(SS_ROWS_QUERY), source rows from
# ss_predict(SS_ROWS_SOURCE) (feature path + target path, SURVEY.md 3.2); the name gather stays in Julia
# (src/core.jl:421).
function _predict_network(A::Network, y::NamedMatrix, ::Type{T}) where {T}
    g = device(A, T)
    qpos = Dict(n => i for (i, n) in enumerate(A.queries))
    spos = Dict(n => i for (i, n) in enumerate(A.sources))
    tpos = Dict(n => i for (i, n) in enumerate(A.targets))
    rows, cols = names(y, 1), names(y, 2)
    tcols = [tpos[c] for c in cols]
    out = zeros(length(rows), length(cols))
    qrows = [(o, qpos[r]) for (o, r) in enumerate(rows) if haskey(qpos, r)]
    srows = [(o, spos[r]) for (o, r) in enumerate(rows) if !haskey(qpos, r)]
    if !isempty(qrows)
        lo, hi = minimum(last, qrows), maximum(last, qrows)
        blk = SimSpreadHIP.predict(g, :query; range=lo:hi)
        for (o, i) in qrows
            out[o, :] = blk[i-lo+1, tcols]
        end
    end
    if !isempty(srows)
        lo, hi = minimum(last, srows), maximum(last, srows)
        blk = SimSpreadHIP.predict(g, :source; range=lo:hi)
        for (o, i) in srows
            out[o, :] = blk[i-lo+1, tcols]
        end
    end
    return NamedArray(out, (rows, cols))
end

_covers(A::Network, y::NamedMatrix) =
    all(c -> c in A.targets, names(y, 2)) && all(r -> (r in A.queries) || (r in A.sources), names(y, 1))

"""
    _node_groups(A, B, y) -> (queries, sources, features, targets) or nothing

Recover the layers of a tri-partite graph from dense named matrices as `construct` lays them out
(src/core.jl:182-198): targets are the columns of `y`; sources the neighbours of targets in `B`; features the
neighbours of sources that are not targets; queries the nodes with an empty row in `B` but not in `A`.  Returns
`nothing` when `A`, `B` are not of that shape (then the general path is taken).
"""
function _node_groups(A::NamedMatrix, B::NamedMatrix, y::NamedMatrix)
    nodes = names(A, 1)
    nodes == names(A, 2) == names(B, 1) == names(B, 2) || return nothing
    pos = Dict(n => i for (i, n) in enumerate(nodes))
    all(t -> haskey(pos, t), names(y, 2)) || return nothing
    Aa, Ba = A.array, B.array
    tset = Set(pos[t] for t in names(y, 2))
    sset = Set(i for i in 1:length(nodes) if !(i in tset) && any(!iszero, Ba[i, collect(tset)]))
    fset = Set(j for j in 1:length(nodes) if !(j in tset) && !(j in sset) && any(i -> Ba[i, j] != 0, sset))
    qset = Set(i for i in 1:length(nodes) if all(iszero, view(Ba, i, :)) && any(!iszero, view(Aa, i, :)))
    isempty(intersect(qset, union(tset, sset, fset))) || return nothing
    grp(s) = [nodes[i] for i in sort(collect(s))]
    q, s, f, t = grp(qset), grp(sset), grp(fset), names(y, 2)
    # everything outside the blocks q-f, s-f, s-t (and their mirror images) must be zero, B = A without the queries
    inblock(i, j) = (i in qset && j in fset) || (i in sset && (j in fset || j in tset))
    for j in 1:length(nodes), i in 1:length(nodes)
        a = Aa[i, j]
        a == Aa[j, i] || return nothing
        a == 0 && continue
        (inblock(i, j) || inblock(j, i)) || return nothing
        expected_b = (i in qset || j in qset) ? 0.0 : a
        Ba[i, j] == expected_b || return nothing
    end
    return q, s, f, t
end

function _predict_general(A::NamedMatrix, B::NamedMatrix, y::NamedMatrix, ::Type{T}) where {T}
    size(A) == size(B) && size(A, 1) == size(A, 2) || throw(DimensionMismatch("A and B must be square matrices of the same size"))
    rpos = Dict(n => i for (i, n) in enumerate(names(A, 1)))
    cpos = Dict(n => i for (i, n) in enumerate(names(A, 2)))
    r = [rpos[n] for n in names(y, 1)]
    c = [cpos[n] for n in names(y, 2)]
    Ba = Matrix{Float64}(B.array)
    g = SimSpreadHIP.graph_general(sparse(Matrix{Float64}(A.array[r, :])), sparse(Ba), sparse(Matrix(Ba[:, c]')); T=T)
    out = SimSpreadHIP.predict(g, :query)
    SimSpreadHIP.destroy!(g)
    return NamedArray(out, (names(y, 1), names(y, 2)))
end

"""
    predict((A, B), ytest; GPU=false)      (src/core.jl:402-423)

The block of `F = A * spread(B)^2` named by the rows and columns of `ytest`, as a `NamedMatrix{Float64}`.
`GPU=true`: fp32 arithmetic, widened (the reference's `CuArray{Float32}` branch, src/core.jl:404,413).
"""
function predict(I::Tuple{Network,Network}, ytest::NamedMatrix; GPU::Bool=false)
    A, B = I
    T = GPU ? Float32 : Float64
    if A.dev === B.dev && B.kind == :B && _covers(A, ytest)
        return _predict_network(A, ytest, T)
    end
    return _predict_general(NamedArray(A), NamedArray(B), ytest, T)
end
function predict(I::Tuple{T,T}, ytest::T; GPU::Bool=false) where {T<:NamedMatrix}
    A, B = I
    P = GPU ? Float32 : Float64
    groups = _node_groups(A, B, ytest)
    if groups !== nothing
        q, s, f, t = groups
        Xq = isempty(q) ? zeros(0, length(f)) : Matrix{Float64}(A[q, f].array)
        Xs, Ys = Matrix{Float64}(A[s, f].array), Matrix{Float64}(A[s, t].array)
        g = SimSpreadHIP.graph(isempty(q) ? nothing : Xq, Xs, Ys; T=P)
        net = Network(:A, q, s, f, t, Xq, Xs, Ys, Dict{DataType,SimSpreadHIP.Graph}(P => g))
        if _covers(net, ytest)
            return _predict_network(net, ytest, P)
        end
    end
    return _predict_general(A, B, ytest, P)
end

"predict(A, B, ytest; GPU=false)   (src/core.jl:424-425)"
predict(A::Network, B::Network, ytest::NamedMatrix; GPU::Bool=false) = predict((A, B), ytest; GPU=GPU)
predict(A::T, B::T, ytest::T; GPU::Bool=false) where {T<:NamedMatrix} = predict((A, B), ytest; GPU=GPU)

"""
    predict(A, ytrain; GPU=false)      (src/core.jl:446-466)

3-layer graph: `W = spread(A)`, returns the sources x targets block (feature path + target path).
"""
function predict(A::Network, ytrain::NamedMatrix; GPU::Bool=false)
    T = GPU ? Float32 : Float64
    if A.kind == :single && _covers(A, ytrain)
        return _predict_network(A, ytrain, T)
    end
    NA = NamedArray(A)
    return _predict_general(NA, NA, ytrain, T)
end
predict(A::T, ytrain::T; GPU::Bool=false) where {T<:NamedMatrix} = _predict_general(A, A, ytrain, GPU ? Float32 : Float64)

# ------------------------------------------------------------------------------------------------ clean!
"clean!(yhat, A, y): column `t` of `yhat` becomes -99 when target `t` has degree 0 in `A` (src/core.jl:478-484)."
function clean!(yhat::NamedArray, A::Network, y::NamedArray)
    _, _, kt = SimSpreadHIP.graph_degrees(device(A, Float64))
    tpos = Dict(n => i for (i, n) in enumerate(A.targets))
    dense = nothing                      # only for names that are not targets of the network (the reference accepts any node)
    for t in names(y, 2)
        deg = if haskey(tpos, t)
            kt[tpos[t]]
        else
            dense === nothing && (dense = Matrix(A))
            k(dense[findfirst(==(t), names(A, 1)), :])
        end
        if deg == 0
            yhat[:, t] .= -99
        end
    end
end
function clean!(yhat::NamedArray, A::NamedArray, y::NamedArray)
    for (t, deg) in zip(names(y, 2), k(A[names(y, 2), :].array))
        if deg == 0
            yhat[:, t] .= -99
        end
    end
end

# ------------------------------------------------------------------------------------------------ cross-validation in one call
"""
    predict_loo(y, X; GPU=false, clean=true)

Every leave-one-out fold `predict(construct(y, X, [s]), y[[s], :])` (+ `clean!`) of the reference's fold loop
(src/core.jl:148-201) from ONE resident graph: row `i` is the fold that holds source `i` out.  `X` is the square
featurized similarity whose column `j` is the feature named after source `j`.
"""
function predict_loo(y::NamedMatrix, X::NamedMatrix; GPU::Bool=false, clean::Bool=true)
    @assert size(y, 1) == size(X, 1) "Labels and features have different number of source nodes"
    g = SimSpreadHIP.graph(nothing, X.array, y.array; T=GPU ? Float32 : Float64)
    out = SimSpreadHIP.predict_loo(g; clean=clean)
    SimSpreadHIP.destroy!(g)
    return NamedArray(out, (names(y, 1), names(y, 2)))
end

"""
    predict_kfold(y, X, groups; GPU=false, clean=true)

All folds of `split(y, k)` (src/core.jl:11-25) in one call: `groups` is the vector of name groups `split` returns;
row `i` is scored with the fold of source `i` held out (members removed from the sources AND their feature columns
dropped, src/core.jl:152-153).
"""
function predict_kfold(y::NamedMatrix, X::NamedMatrix, groups::AbstractVector; GPU::Bool=false, clean::Bool=true)
    @assert size(y, 1) == size(X, 1) "Labels and features have different number of source nodes"
    fold_of = Dict{String,Int}()
    for (fi, members) in enumerate(groups), m in members
        fold_of[string(m)] = fi
    end
    folds = [fold_of[n] for n in names(y, 1)]
    g = SimSpreadHIP.graph(nothing, X.array, y.array; T=GPU ? Float32 : Float64)
    out = SimSpreadHIP.predict_kfold(g, folds; clean=clean)
    SimSpreadHIP.destroy!(g)
    return NamedArray(out, (names(y, 1), names(y, 2)))
end

end # module
