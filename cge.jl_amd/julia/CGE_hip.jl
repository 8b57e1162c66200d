# CGE_hip.jl -- the Julia side of the drop-in: same exported names and positional signatures as
# CGE.jl's hot path (src/CGE.jl:11-21), each a thin `ccall` into libcge_hip.so (include/cge_hip.h).
#
# UNTESTED -- NEVER EXECUTED IN THIS REPOSITORY'S CI: the build image has no `julia` binary (DESIGN.md, "Boundary").
# The Python mirror (cge.jl_amd/api.py) binds the identical C-ABI and is what the parity tests drive;
# this file is the stub a CGE.jl maintainer would add.  Julia arrays are passed as they are: Int64
# 1-based ids, column-major matrices -- the C-ABI was laid out for exactly that.
module CGE_hip

export landmarks, wGCL, wGCL_directed, split_cluster_rss, split_cluster_rss2, split_cluster_size,
       split_cluster_diameter

const LIB = get(ENV, "CGE_HIP_LIB", joinpath(@__DIR__, "..", "csrc", "build", "libcge_hip.so"))

# The reference passes the split rule as a function (src/auxilary.jl:64-67); the C-ABI takes an enum.  The rule
# is mapped BY NAME, so `CGE.split_cluster_rss` (what CGE's own `parseargs` returns) and the placeholders below
# both work.  Use `using CGE: parseargs` next to `using CGE_hip` (a plain `using CGE` would clash on the exported
# `landmarks`, `wGCL`, `wGCL_directed`).
split_cluster_rss() = nothing
split_cluster_rss2() = nothing
split_cluster_size() = nothing
split_cluster_diameter() = nothing
const METHOD_CODE = Dict{String,Cint}("split_cluster_rss" => 0, "split_cluster_rss2" => 1,
                                      "split_cluster_size" => 2, "split_cluster_diameter" => 3)
function method_code(method::Function)
    name = string(nameof(method))
    haskey(METHOD_CODE, name) || throw(ArgumentError("unknown split rule $name (expected one of $(collect(keys(METHOD_CODE))))"))
    return METHOD_CODE[name]
end

mutable struct Ctx
    h::Ptr{Cvoid}
    function Ctx(device::Integer = 0)
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:cge_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Cint, Ptr{Cvoid}), ref, device, C_NULL)
        rc == 0 || error("cge_create failed with code $rc (no MI355X visible? there is no CPU fallback)")
        c = new(ref[])
        finalizer(x -> (x.h != C_NULL && ccall((:cge_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.h); x.h = C_NULL), c)
        return c
    end
end

const DEFAULT = Ref{Union{Nothing,Ctx}}(nothing)
function ctx()
    if DEFAULT[] === nothing
        DEFAULT[] = Ctx()
        # tear the context down while the HIP runtime is still alive (streams and events must not outlive it)
        atexit(() -> (c = DEFAULT[]; c !== nothing && c.h != C_NULL &&
                      (ccall((:cge_destroy, LIB), Cvoid, (Ptr{Cvoid},), c.h); c.h = C_NULL)))
    end
    return DEFAULT[]
end

lasterr(c::Ctx) = unsafe_string(ccall((:cge_last_error, LIB), Cstring, (Ptr{Cvoid},), c.h))

# status code -> the exception the reference would have thrown
function check(c::Ctx, rc::Cint)
    rc == 0 && return
    msg = lasterr(c)
    rc == -1 && throw(AssertionError(msg))                 # @assert (src/divergence.jl:50,81,303,363)
    rc == -2 && throw(ErrorException("Trying to split homogenous cluster"))       # src/landmarks.jl:166
    rc == -3 && throw(ErrorException("Unexpected empty cluster generated"))       # src/landmarks.jl:298
    error("libcge_hip error $rc: $msg")
end

function set_inputs!(c::Ctx, edges::Matrix{Int}, weights::Vector{Float64}, vweights::Vector{Float64},
                     comm::Matrix{Int}, embedding::Matrix{Float64})
    m, n, d = size(edges, 1), size(embedding, 1), size(embedding, 2)
    src, dst = pointer(edges), pointer(edges, m + 1)      # the two columns of the m x 2 matrix
    GC.@preserve edges weights vweights comm embedding begin
        check(c, ccall((:cge_set_graph, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Int64, Int64),
                       c.h, src, dst, weights, m, n))
        check(c, ccall((:cge_set_embedding, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64), c.h, embedding, n, d))
        check(c, ccall((:cge_set_vertex_data, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}, Int64),
                       c.h, comm, vweights, n))
    end
end

"""
    landmarks(edges, weights, vweights, clusters, comm, embedding, verbose, land, forced, method, directed)

Drop-in for `CGE.landmarks` (src/landmarks.jl:365-367); returns the same 7-tuple (:465).
"""
function landmarks(edges::Array{Int,2}, weights::Vector{Float64}, vweights::Vector{Float64},
                   clusters::Vector{Vector{Int}}, comm::Array{Int,2}, embedding::Array{Float64,2},
                   verbose::Bool, land::Int, forced::Int, method::Function, directed::Bool)
    c = ctx()
    verbose && println("Starts landmark generation")
    set_inputs!(c, edges, weights, vweights, comm, embedding)
    flat = reduce(vcat, clusters; init = Int[])
    off = Int64[0; cumsum(length.(clusters))]
    N, ne, trunc = Ref{Int64}(0), Ref{Int64}(0), Ref{Cint}(0)
    check(c, ccall((:cge_landmarks_run, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Int64, Int64, Int64, Cint, Cint, Ref{Int64}, Ref{Int64}, Ref{Cint}),
                   c.h, flat, off, length(clusters), land, forced, method_code(method), directed, N, ne, trunc))
    trunc[] != 0 && @warn "Requested number of clusters larger than unique no. embeddings. Truncating to $(N[]) landmarks."
    verbose && println("Landmarks generated"); verbose && println("Using $(N[]) landmarks")
    n, d = size(embedding)
    dii, embed, cluster = zeros(N[]), zeros(N[], d), zeros(Int, N[])
    ledges, lw, lweight, v_to_l = zeros(Int, ne[], 2), zeros(ne[]), zeros(N[]), zeros(Int, n)
    check(c, ccall((:cge_landmarks_fetch, LIB), Cint,
                   (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}),
                   c.h, dii, embed, cluster, ledges, lw, lweight, v_to_l))
    return dii, embed, reshape(cluster, :, 1), ledges, lw, lweight, v_to_l
end

# mirrors `cge_trace` (include/cge_hip.h): n_alpha, iters[64], div[64], auc[64]
mutable struct Trace
    n_alpha::Int64
    iters::NTuple{64,Int64}
    div::NTuple{64,Float64}
    auc::NTuple{64,Float64}
    Trace() = new(0, ntuple(_ -> 0, 64), ntuple(_ -> 0.0, 64), ntuple(_ -> 0.0, 64))
end

# mirrors `cge_wgcl_args` (include/cge_hip.h) field for field
struct WgclArgs
    edges_src::Ptr{Int64}; edges_dst::Ptr{Int64}; eweights::Ptr{Float64}; m::Int64
    comm::Ptr{Int64}; n_comm::Int64
    embed::Ptr{Float64}; embed_rows::Int64; d::Int64
    distances::Ptr{Float64}; n_distances::Int64
    vweights::Ptr{Float64}
    init_vweights::Ptr{Float64}; n_init::Int64
    v_to_l::Ptr{Int64}; n_v_to_l::Int64
    init_edges_src::Ptr{Int64}; init_edges_dst::Ptr{Int64}; m_init::Int64
    init_eweights::Ptr{Float64}; init_embed::Ptr{Float64}
    split::Cint; seed::Int64; auc_samples::Int64; verbose::Cint; directed::Cint
    pos_idx::Ptr{Int64}; neg_i::Ptr{Int64}; neg_j::Ptr{Int64}; pos_idx2::Ptr{Int64}; n_sample_sets::Int64
end

function _wgcl(directed::Bool, edges, eweights, comm, embed, distances, vweights, init_vweights, v_to_l,
               init_edges, init_eweights, init_embed, split, seed, auc_samples, verbose)
    c = ctx()
    m, mi = size(edges, 1), size(init_edges, 1)
    if verbose                                           # the reference's own lines (src/divergence.jl:43-52,75)
        println("auc_samples: $auc_samples")
        println("Graph has $(maximum(edges)) vertices and $m edges")
        !isempty(v_to_l) && mi > 0 && println("Original graph has $(maximum(init_edges)) vertices and $mi edges")
        println("Graph has $(maximum(comm)) communities")
        println("Embedding has $(size(embed, 2)) dimensions")
    end
    out, olen = zeros(7), Ref{Cint}(7)
    GC.@preserve edges eweights comm embed distances vweights init_vweights v_to_l init_edges init_eweights init_embed begin
        a = WgclArgs(pointer(edges), pointer(edges, m + 1), pointer(eweights), m,
                     pointer(comm), size(comm, 1), pointer(embed), size(embed, 1), size(embed, 2),
                     pointer(distances), length(distances), pointer(vweights),
                     pointer(init_vweights), length(init_vweights), pointer(v_to_l), length(v_to_l),
                     mi > 0 ? pointer(init_edges) : C_NULL, mi > 0 ? pointer(init_edges, mi + 1) : C_NULL, mi,
                     pointer(init_eweights), isempty(init_embed) ? C_NULL : pointer(init_embed),
                     split, seed, auc_samples, verbose, directed,
                     C_NULL, C_NULL, C_NULL, C_NULL, 0)     # samples: drawn by the library (pass arrays to own the RNG)
        tr = Trace()
        rc = ccall((:cge_wgcl, LIB), Cint, (Ptr{Cvoid}, Ref{WgclArgs}, Ptr{Float64}, Ref{Cint}, Ref{Trace}),
                   c.h, Ref(a), out, olen, tr)
        check(c, rc)
    end
    write(stderr, "."^tr.n_alpha, "\n")                  # one "." per alpha (src/divergence.jl:140) and the newline (:255)
    return out[1:olen[]]
end

"Drop-in for `CGE.wGCL` (src/divergence.jl:27-31)."
wGCL(edges::Array{Int,2}, eweights::Vector{Float64}, comm::Matrix{Int}, embed::Matrix{Float64},
     distances::Vector{Float64}, vweights::Vector{Float64}, init_vweights::Vector{Float64}, v_to_l::Vector{Int},
     init_edges::Array{Int,2}, init_eweights::Vector{Float64}, init_embed::Matrix{Float64}, split::Bool,
     seed::Int = -1, auc_samples::Int = 10000, verbose::Bool = false) =
    _wgcl(false, edges, eweights, comm, embed, distances, vweights, init_vweights, v_to_l, init_edges,
          init_eweights, init_embed, split, seed, auc_samples, verbose)

"Drop-in for `CGE.wGCL_directed` (src/divergence.jl:282-286)."
wGCL_directed(edges::Array{Int,2}, eweights::Vector{Float64}, comm::Matrix{Int}, embed::Matrix{Float64},
              distances::Vector{Float64}, vweights::Vector{Float64}, init_vweights::Vector{Float64},
              v_to_l::Vector{Int}, init_edges::Array{Int,2}, init_eweights::Vector{Float64},
              init_embed::Matrix{Float64}, split::Bool, seed::Int = -1, auc_samples::Int = 10000,
              verbose::Bool = false) =
    _wgcl(true, edges, eweights, comm, embed, distances, vweights, init_vweights, v_to_l, init_edges,
          init_eweights, init_embed, split, seed, auc_samples, verbose)

"""
    read_table(fn) -> Matrix{Float64}

Stand-in for `readdlm(fn, Float64)` in `parseargs` (src/auxilary.jl:80-168): parallel reader of the library, header
line (node2vec's "n d") skipped as the reference's retry with `skipstart = 1` does.  Needs no GPU.
"""
function read_table(fn::AbstractString; threads::Integer = 0)
    rows, cols, hdr, h = Ref{Int64}(0), Ref{Int64}(0), Ref{Cint}(0), Ref{Ptr{Cvoid}}(C_NULL)
    err = zeros(UInt8, 512)
    rc = ccall((:cge_text_table_open, LIB), Cint,
               (Cstring, Cint, Ref{Int64}, Ref{Int64}, Ref{Cint}, Ref{Ptr{Cvoid}}, Ptr{UInt8}, Int64),
               fn, threads, rows, cols, hdr, h, err, length(err))
    rc == 0 || throw(ArgumentError(unsafe_string(pointer(err))))
    M = Matrix{Float64}(undef, rows[], cols[])
    rc = ccall((:cge_text_table_parse, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint, Ptr{UInt8}, Int64),
               h[], M, 1, err, length(err))
    ccall((:cge_text_table_close, LIB), Cvoid, (Ptr{Cvoid},), h[])
    rc == 0 || throw(ArgumentError(unsafe_string(pointer(err))))
    return M
end

end # module
