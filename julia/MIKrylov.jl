# MIKrylov.jl -- Julia shim that makes libexpv_mi.so (the MI355X Krylov exp(tA)v engine) a drop-in for the Krylov
# path of SciML/ExponentialUtilities.jl: device-resident array / operator / KrylovSubspace types, and methods of the
# reference's own generic functions (arnoldi!, lanczos!, expv!, phiv!, _phiv!, phiv_timestep!, expv_timestep!, kiops,
# _phiv_timestep_caches, the error-estimate expv!) that forward to the C ABI of include/expv_mi.h with `ccall`.
# OrdinaryDiffEq's exponential integrators keep calling the reference API; dispatch on MIVector / MIOperator picks
# these methods.
#
# STATUS: source only.  The build image has no Julia, so this file has never been executed; what IS checked here:
#   * method signatures were checked by hand against the reference's for dispatch ambiguities (a method that is more specific in
#     one argument and less in another is a MethodError at the first call): expv! is split into `t::Real` / `t::Complex` like
#     krylov_phiv.jl:200-203, :252-255 for that reason,
#   * every `ccall` below names an exported symbol with the argument order of include/expv_mi.h
#     (tests/test_abi_cpu.py::test_julia_shim_calls_match_the_header),
#   * the option structs restated below have the library's field order and types
#     (tests/test_abi_cpu.py::test_struct_layouts_match_the_library; at load time `check_abi()` repeats that check
#     against expv_mi_abi_sizeof, so a mismatched library refuses to load),
#   * the identical entry points with the identical marshalling are exercised by the Python/ctypes mirror
#     (exponentialutilities.jl_amd/api.py) in the -m gpu parity tests.
#
# Reference methods replaced (all under /root/reference/src/):
#   arnoldi.jl:50-93,161-180,345-377,456-490   KrylovSubspace storage, arnoldi, arnoldi!, lanczos!
#   krylov_phiv.jl:125-168,200-280,563-653       expv, expv!, phiv, phiv!, _phiv!
#   krylov_phiv_adaptive.jl:57-114,184-232,260-453,502-511   expv_timestep(!), phiv_timestep(!), _phiv_timestep_caches
#   kiops.jl:57-281                               kiops
#   krylov_phiv_error_estimate.jl:96-101,149-207  get_subspace_cache, expv!(w,t,A,b,Ks,cache)
module MIKrylov

using LinearAlgebra, SparseArrays
import ExponentialUtilities
import ExponentialUtilities: KrylovSubspace, arnoldi, arnoldi!, lanczos!, expv, expv!, phiv, phiv!, _phiv!,
                             phiv_timestep, phiv_timestep!, expv_timestep, expv_timestep!, kiops,
                             _phiv_timestep_caches, get_subspace_cache, getV, getH

const lib = get(ENV, "EXPV_MI_LIB", joinpath(@__DIR__, "..", "exponentialutilities.jl_amd", "libexpv_mi.so"))
const F64, C64, F32, C32 = Cint(0), Cint(1), Cint(2), Cint(3)
const HOST, DEVICE = Cint(0), Cint(1)
# every BlasFloat (ExponentialUtilities.jl:19): the 32-bit types run natively on 32-bit storage (two-kernel step, fp64 projection
# sums); kiops is Float64-only like the reference method and answers Unsupported for them
const MIScalar = Union{Float64, ComplexF64, Float32, ComplexF32}
dtype(::Type{Float64}) = F64
dtype(::Type{ComplexF64}) = C64
dtype(::Type{Float32}) = F32
dtype(::Type{ComplexF32}) = C32

# ---- option / result structs: field order and types of include/expv_mi.h (verified by check_abi) ----------------
struct ArnoldiOpts
    m::Cint
    iop::Cint
    init::Cint
    ishermitian::Cint
    ortho::Cint
    flags::Cint
    tol::Cdouble
end
struct ExpvStats
    m_used::Cint
    wasbreakdown::Cint
    matvecs::Cint
    path_flags::Cint
    beta::Cdouble
end
struct TimestepOpts
    tau::Cdouble
    tol::Cdouble
    delta::Cdouble
    gamma::Cdouble
    opnorm::Cdouble
    has_opnorm::Cint
    m::Cint
    iop::Cint
    correct::Cint
    adaptive::Cint
    ishermitian::Cint
    verbose::Cint
    ortho::Cint
    no_basis_reuse::Cint
    reserved::Cint
    NA::Int64
    print::Ptr{Cvoid}
    print_user::Ptr{Cvoid}
end
struct TimestepStats
    num_timesteps::Cint
    matvecs::Cint
    m_final::Cint
    arnoldi_calls::Cint
    arnoldi_reused::Cint
    stalled_steps::Cint
end
struct KiopsOpts
    mmin::Cint
    mmax::Cint
    m::Cint
    iop::Cint
    ishermitian::Cint
    task1::Cint
    ortho::Cint
    reserved::Cint
    tol::Cdouble
end
function check_abi()
    for (kind, T) in ((0, ArnoldiOpts), (1, ExpvStats), (2, TimestepOpts), (3, TimestepStats), (4, KiopsOpts))
        want = ccall((:expv_mi_abi_sizeof, lib), Csize_t, (Cint,), kind)
        want == sizeof(T) || error("libexpv_mi.so: $(T) is $(want) bytes in the library, $(sizeof(T)) here")
    end
end

# ---- context: one GPU + one HIP stream --------------------------------------------------------------------------
mutable struct Ctx
    h::Ptr{Cvoid}
end
function Ctx(device::Integer = 0)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:expv_mi_ctx_create, lib), Cint, (Cint, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, r), C_NULL)
    c = Ctx(r[])
    # Julia runs finalizers in no particular order at exit: the library expects that -- expv_mi_ctx_destroy clears the back
    # pointers of the operator / KrylovSubspace / cache handles that outlive it, and their own destroy then only frees their
    # device memory (tests/test_gpu_configs.py: test_handles_may_outlive_their_context...; tests/c_harness does exactly this)
    finalizer(c -> (ccall((:expv_mi_ctx_destroy, lib), Cint, (Ptr{Cvoid},), c.h); c.h = C_NULL), c)
    c
end
const CTX = Ref{Ctx}()
ctx() = (isassigned(CTX) || (check_abi(); CTX[] = Ctx()); CTX[])
sync() = check(ccall((:expv_mi_ctx_sync, lib), Cint, (Ptr{Cvoid},), ctx().h), ctx().h)
# engine options of the context by name ("pipeline", "wave", "fused", "dia", "mailbox", "nontemporal", "stencil", ...: expv_mi.h)
set_option!(name::AbstractString, value::Integer) =
    check(ccall((:expv_mi_ctx_set_option, lib), Cint, (Ptr{Cvoid}, Cstring, Int64), ctx().h, name, value), ctx().h)
function get_option(name::AbstractString)
    v = Ref{Int64}(0)
    check(ccall((:expv_mi_ctx_get_option, lib), Cint, (Ptr{Cvoid}, Cstring, Ref{Int64}), ctx().h, name, v), ctx().h)
    v[]
end
# cumulative counters: Krylov steps, factorisations, on the single-pass step, overlapped, redone serially, redone without the
# wave form, operator applications outside a factorisation, reserved
function counters()
    out = zeros(Int64, 8)
    check(ccall((:expv_mi_ctx_counters, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}), ctx().h, out), ctx().h)
    out
end

# device self-test of the kernels' cross-lane sums (mismatching lanes per class; all zero when healthy)
function selftest()
    out = zeros(Int64, 8)
    check(ccall((:expv_mi_ctx_selftest, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}), ctx().h, out), ctx().h)
    out
end

function check(code::Integer, h)
    code == 0 && return nothing
    msg = unsafe_string(ccall((:expv_mi_last_error, lib), Cstring, (Ptr{Cvoid},), h))
    code == 1 && throw(DimensionMismatch(msg))
    # (where the reference's `ceil(Int, x)` meets a non-finite x -- kiops.jl:210, krylov_phiv_adaptive.jl:470 -- it throws InexactError)
    code == 2 && occursin("InexactError", msg) && throw(InexactError(:ceil, Int, NaN))
    code == 2 && throw(ArgumentError(msg))
    code == 3 && throw(AssertionError(msg))
    code == 4 && throw(LinearAlgebra.SingularException(0))
    code == 6 && throw(OutOfMemoryError())
    code == 8 && throw(BoundsError())
    error("expv_mi status $code: $msg")          # 5 (error(...) in the reference), 7 (HIP)
end

# ---- device arrays (column-major, like Array) ---------------------------------------------------------------------
mutable struct MIArray{T, N} <: AbstractArray{T, N}
    ptr::Ptr{Cvoid}
    dims::NTuple{N, Int}
    owned::Bool
end
const MIVector{T} = MIArray{T, 1}
const MIMatrix{T} = MIArray{T, 2}
const MIVecOrMat{T} = Union{MIVector{T}, MIMatrix{T}}
Base.size(a::MIArray) = a.dims
Base.IndexStyle(::Type{<:MIArray}) = IndexLinear()
Base.getindex(a::MIArray, i...) = error("MIArray lives in HBM: copy it to the host with Array(a) first")
Base.show(io::IO, a::MIArray{T}) where {T} = print(io, join(a.dims, "x"), " MIArray{", T, "} in HBM @", a.ptr)      # (the generic show would index)
Base.show(io::IO, ::MIME"text/plain", a::MIArray) = show(io, a)
function MIArray{T}(::UndefInitializer, dims::Vararg{Int, N}) where {T <: MIScalar, N}
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:expv_mi_malloc, lib), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), ctx().h, max(prod(dims), 1) * sizeof(T), r), ctx().h)
    a = MIArray{T, N}(r[], dims, true)
    # (never ctx() here: a finalizer must not create a context, and the context's own finalizer may have run already --
    #  expv_mi_free does not dereference the handle it is given)
    finalizer(a -> a.owned && ccall((:expv_mi_free, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), isassigned(CTX) ? CTX[].h : C_NULL, a.ptr), a)
    a
end
MIArray{T, N}(::UndefInitializer, dims::Vararg{Int, N}) where {T, N} = MIArray{T}(undef, dims...)      # VType(undef, rows, cols), arnoldi.jl:68
Base.similar(a::MIArray, ::Type{T}, dims::Dims) where {T} = MIArray{T}(undef, dims...)                # similar(b, T, (n, m+1)), arnoldi.jl:171
Base.similar(a::MIArray{T}, dims::Dims) where {T} = MIArray{T}(undef, dims...)
function MIArray(x::Array{T}) where {T <: MIScalar}
    d = MIArray{T}(undef, size(x)...)
    check(ccall((:expv_mi_memcpy_h2d, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{T}, Csize_t), ctx().h, d.ptr, x, sizeof(x)), ctx().h)
    d
end
function Base.Array(d::MIArray{T}) where {T}
    x = Array{T}(undef, d.dims)
    check(ccall((:expv_mi_memcpy_d2h, lib), Cint, (Ptr{Cvoid}, Ptr{T}, Ptr{Cvoid}, Csize_t), ctx().h, x, d.ptr, sizeof(x)), ctx().h)
    x
end
colptr(a::MIMatrix{T}, j::Integer) where {T} = a.ptr + (j - 1) * a.dims[1] * sizeof(T)
ld(a::MIVector) = max(a.dims[1], 1)
ld(a::MIMatrix) = max(a.dims[1], 1)
ncols(a::MIVector) = 1
ncols(a::MIMatrix) = a.dims[2]

# ---- operator: the contract of docs/src/interfaces.md:7-36 (eltype, size, mul!, ishermitian, opnorm) --------------
mutable struct MIOperator{T}
    h::Ptr{Cvoid}
    n::Int
    herm::Bool
    nnz::Int
    opnorm_inf::Float64
end
function wrap_operator(::Type{T}, h::Ptr{Cvoid}) where {T}
    n, nz, hm, on, dt = Ref{Int64}(0), Ref{Int64}(0), Ref{Cint}(0), Ref{Cdouble}(0), Ref{Cint}(0)
    check(ccall((:expv_mi_op_info, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Cint}, Ref{Cdouble}, Ref{Cint}), h, n, nz, hm, on, dt), ctx().h)
    op = MIOperator{T}(h, n[], hm[] != 0, nz[], on[])
    finalizer(o -> ccall((:expv_mi_op_destroy, lib), Cint, (Ptr{Cvoid},), o.h), op)
    op
end
function MIOperator(A::SparseMatrixCSC{T, Int64}) where {T <: MIScalar}          # Julia's own layout, 1-based
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:expv_mi_op_create_csc, lib), Cint,
                (Ptr{Cvoid}, Cint, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{T}, Cint, Ref{Ptr{Cvoid}}),
                ctx().h, dtype(T), size(A, 1), A.colptr, A.rowval, A.nzval, 1, r), ctx().h)
    wrap_operator(T, r[])
end
# A refreshed in place on the same pattern (a Jacobian updated every step): refill the uploaded operator instead of building
# a new one -- the reference reads A at call time, this is how a caller tells the device copy.  ~10x cheaper than MIOperator(A).
function update_values!(op::MIOperator{T}, A::SparseMatrixCSC{T, Int64}) where {T <: MIScalar}
    length(A.nzval) == op.nnz && size(A, 1) == op.n || throw(DimensionMismatch("update_values!: same pattern as at creation required"))
    check(ccall((:expv_mi_op_update_values, lib), Cint, (Ptr{Cvoid}, Ptr{T}, Cint), op.h, A.nzval, HOST), ctx().h)
    n, nz, hm, on, dt = Ref{Int64}(0), Ref{Int64}(0), Ref{Cint}(0), Ref{Cdouble}(0), Ref{Cint}(0)
    check(ccall((:expv_mi_op_info, lib), Cint, (Ptr{Cvoid}, Ref{Int64}, Ref{Int64}, Ref{Cint}, Ref{Cdouble}, Ref{Cint}), op.h, n, nz, hm, on, dt), ctx().h)
    op.herm = hm[] != 0
    op.opnorm_inf = on[]
    op
end
# How the library stores a sparse operator: reordered = true when it kept P A P' (P = reverse Cuthill-McKee, context option
# "reorder") because that puts an unstructured operator on the single-pass Krylov step.  Nothing changes for the caller: vectors
# go in and come out in A's own ordering, H / beta / results are those of arnoldi(A, b) (rounding apart).
function reorder_info(op::MIOperator)
    out = zeros(Int64, 4)
    check(ccall((:expv_mi_op_reorder_info, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}), op.h, out), ctx().h)
    (reordered = out[1] != 0, bandwidth_before = out[2], bandwidth_after = out[3], setup_s = 1.0e-6 * out[4])
end
# ordering plans by pattern (process-wide): re-creating an operator with a pattern seen before reuses its row ordering / patch plan
function plan_cache(; clear::Bool = false, capacity::Union{Nothing, Integer} = nothing)
    out = zeros(Int64, 4)
    clear && check(ccall((:expv_mi_plan_cache, lib), Cint, (Cint, Int64, Ptr{Int64}), 1, 0, out), C_NULL)
    capacity === nothing || check(ccall((:expv_mi_plan_cache, lib), Cint, (Cint, Int64, Ptr{Int64}), 2, capacity, out), C_NULL)
    check(ccall((:expv_mi_plan_cache, lib), Cint, (Cint, Int64, Ptr{Int64}), 0, 0, out), C_NULL)
    (plans = out[1], hits = out[2], misses = out[3], capacity = out[4])
end
# 2-D grid stencils (context option "patch", default on): stored in a grid-patch ordering -- a special case of the reordering above,
# equally invisible to the caller.  patch_form, the row length of the recognised grid, tiles, longest / mean ring of a tile.
function patch_info(op::MIOperator)
    out = zeros(Int64, 8)
    check(ccall((:expv_mi_op_patch_info, lib), Cint, (Ptr{Cvoid}, Ptr{Int64}), op.h, out), ctx().h)
    (patch_form = out[1] != 0, grid_row_length = out[2], tiles = out[3], longest_ring = out[4], mean_ring = out[3] > 0 ? out[5] / out[3] : 0.0)
end
MIOperator(A::SparseMatrixCSC{T}) where {T <: MIScalar} = MIOperator(SparseMatrixCSC{T, Int64}(A))      # (other index types: converted once)
function MIOperator(A::Matrix{T}) where {T <: MIScalar}
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:expv_mi_op_create_dense, lib), Cint, (Ptr{Cvoid}, Cint, Int64, Ptr{T}, Int64, Cint, Ref{Ptr{Cvoid}}),
                ctx().h, dtype(T), size(A, 1), A, max(size(A, 1), 1), HOST, r), ctx().h)
    wrap_operator(T, r[])
end
function MIOperator(A::MIMatrix{T}) where {T}                                    # already in HBM (the caller keeps it alive)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:expv_mi_op_create_dense, lib), Cint, (Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Int64, Cint, Ref{Ptr{Cvoid}}),
                ctx().h, dtype(T), size(A, 1), A.ptr, ld(A), DEVICE, r), ctx().h)
    wrap_operator(T, r[])
end
# Matrix-free operator: anything that implements the reference's operator contract (docs/src/interfaces.md:7-36, exercised by
# test/basictests.jl:786-816): eltype, size, LinearAlgebra.mul!(y, A, x) on device vectors, ishermitian.  The library calls
# back with device pointers on its stream; the Julia object is kept alive by the operator that wraps it.
struct MatVecBox
    A::Any
    T::DataType
    n::Int
end
const MATVEC_ROOTS = IdDict{Ptr{Cvoid}, Any}()                # callback user pointer -> boxed operator (rooted while the handle lives)
function matvec_trampoline(user::Ptr{Cvoid}, xp::Ptr{Cvoid}, yp::Ptr{Cvoid}, stream::Ptr{Cvoid})::Cint
    try
        box = unsafe_pointer_to_objref(user)::Base.RefValue{MatVecBox}
        b = box[]
        x = MIArray{b.T, 1}(xp, (b.n,), false)
        y = MIArray{b.T, 1}(yp, (b.n,), false)
        LinearAlgebra.mul!(y, b.A, x)                          # the user's method, on MIVector arguments
        return Cint(0)
    catch
        return Cint(1)                                         # surfaces as ArgumentError from the library call
    end
end
function MIOperator(A, ::Type{T} = eltype(A); ishermitian::Bool = LinearAlgebra.ishermitian(A), nnz_hint::Integer = 0) where {T <: MIScalar}
    n = size(A, 1)
    box = Ref(MatVecBox(A, T, n))
    user = pointer_from_objref(box)
    fn = @cfunction(matvec_trampoline, Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}))
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:expv_mi_op_create_callback, lib), Cint,
                (Ptr{Cvoid}, Cint, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Int64, Ref{Ptr{Cvoid}}),
                ctx().h, dtype(T), n, fn, user, ishermitian ? 1 : 0, nnz_hint, r), ctx().h)
    MATVEC_ROOTS[r[]] = box
    op = wrap_operator(T, r[])
    finalizer(o -> delete!(MATVEC_ROOTS, o.h), op)
    op
end
Base.eltype(::MIOperator{T}) where {T} = T
Base.size(A::MIOperator) = (A.n, A.n)
Base.size(A::MIOperator, d::Integer) = d <= 2 ? A.n : 1
LinearAlgebra.ishermitian(A::MIOperator) = A.herm
LinearAlgebra.opnorm(A::MIOperator, p::Real = Inf) = p == Inf ? A.opnorm_inf : error("MIOperator: only opnorm(A, Inf) is kept")
SparseArrays.nnz(A::MIOperator) = A.nnz
function LinearAlgebra.mul!(y::MIVector{T}, A::MIOperator{T}, x::MIVector{T}) where {T}                 # arnoldi.jl:185
    check(ccall((:expv_mi_op_apply, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint), A.h, x.ptr, DEVICE, y.ptr, DEVICE), ctx().h)
    y
end

# ---- KrylovSubspace on the device ----------------------------------------------------------------------------------
# Ks.V is an MIMatrix over library-owned HBM; Ks.H stays a host Matrix{U} (arnoldi.jl:69) that WRAPS the library's host
# buffer (unsafe_wrap), so host code that edits H in place (kiops.jl:149-160, OrdinaryDiffEq reading Ks.H) keeps working.
mutable struct MIHandle
    h::Ptr{Cvoid}
    rows::Int            # n + augmented (Ks.V is padded: size(Ks.V, 1) is the leading dimension)
end
const HANDLES = WeakKeyDict{Any, MIHandle}()
handle(Ks) = HANDLES[Ks].h
function mi_subspace(::Type{T}, ::Type{U}, n::Integer, maxiter::Integer = 30, augmented::Integer = 0) where {T, U}
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:expv_mi_ks_create, lib), Cint, (Ptr{Cvoid}, Cint, Cint, Int64, Cint, Cint, Ref{Ptr{Cvoid}}),
                ctx().h, dtype(T), dtype(U), n, maxiter, augmented, r), ctx().h)
    V, H = views_of(T, U, r[], n + augmented)
    Ks = KrylovSubspace{T, U, real(T), typeof(V), Matrix{U}}(maxiter, maxiter, augmented, zero(real(T)), false, V, H)
    hd = MIHandle(r[], n + augmented)
    finalizer(hd -> ccall((:expv_mi_ks_destroy, lib), Cint, (Ptr{Cvoid},), hd.h), hd)
    HANDLES[Ks] = hd
    Ks
end
function views_of(::Type{T}, ::Type{U}, h::Ptr{Cvoid}, rows::Integer) where {T, U}
    hp, ldh, nr, nc = Ref{Ptr{Cvoid}}(C_NULL), Ref{Cint}(0), Ref{Cint}(0), Ref{Cint}(0)
    check(ccall((:expv_mi_ks_H, lib), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Cint}, Ref{Cint}, Ref{Cint}), h, hp, ldh, nr, nc), ctx().h)
    H = unsafe_wrap(Array, Ptr{U}(hp[]), (Int(ldh[]), Int(nc[])))
    vp, ldv = Ref{Ptr{Cvoid}}(C_NULL), Ref{Int64}(0)
    check(ccall((:expv_mi_ks_V_devptr, lib), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Int64}), h, vp, ldv), ctx().h)
    V = MIArray{T, 2}(vp[], (Int(ldv[]), size(H, 1)), false)       # library-owned: no finalizer; ldv >= rows (padded)
    V, H
end
# KrylovSubspace{T,U,MIMatrix{T}}(n, maxiter, augmented): the reference's own constructor form (arnoldi.jl:63-73)
KrylovSubspace{T, U, MIMatrix{T}}(n::Integer, maxiter::Integer = 30, augmented::Integer = false) where {T, U} =
    mi_subspace(T, U, n, maxiter, Int(augmented))
function sync_fields!(Ks)
    m, mi, aug, beta, wb = Ref{Cint}(0), Ref{Cint}(0), Ref{Cint}(0), Ref{Cdouble}(0), Ref{Cint}(0)
    check(ccall((:expv_mi_ks_get, lib), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ref{Cint}, Ref{Cdouble}, Ref{Cint}),
                handle(Ks), m, mi, aug, beta, wb), ctx().h)
    Ks.m, Ks.beta, Ks.wasbreakdown = m[], beta[], wb[] != 0
    Ks
end
const MIKs{T, U} = KrylovSubspace{T, U, <:Any, <:MIMatrix}
function Base.resize!(Ks::MIKs{T, U}, maxiter::Integer) where {T, U}                                     # arnoldi.jl:80-93
    check(ccall((:expv_mi_ks_resize, lib), Cint, (Ptr{Cvoid}, Cint), handle(Ks), maxiter), ctx().h)
    Ks.V, Ks.H = views_of(T, U, handle(Ks), size(Ks.V, 1))        # the library reallocated both
    Ks.maxiter = maxiter
    sync_fields!(Ks)
end
# getV hands the caller an orthonormal basis: the library normalises lazily (expv_mi_ks_V_devptr materialises it)
function getV(Ks::MIKs{T, U}) where {T, U}
    vp, ldv = Ref{Ptr{Cvoid}}(C_NULL), Ref{Int64}(0)
    check(ccall((:expv_mi_ks_V_devptr, lib), Cint, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Int64}), handle(Ks), vp, ldv), ctx().h)
    MIArray{T, 2}(vp[], (Int(ldv[]), Ks.m + 1), false)       # leading dimension ldv >= rows; rows beyond n + augmented are zero padding
end

opts(m, tol, iop, init, herm; flags = 0) = Ref(ArnoldiOpts(m, iop, init, herm, 0, flags, tol))
const DEFER_TAIL = Cint(1)      # EXPV_MI_ARNOLDI_DEFER_TAIL: arnoldi! / lanczos! return before the closing pass has finished; every accessor below
                                # (getH, getV, Ks.m, expv!, phiv!, ...) goes through a library call that collects it

# arnoldi!(Ks, A, b; tol, m, ishermitian, opnorm, iop, init)                                  (src/arnoldi.jl:345-377)
function arnoldi!(Ks::MIKs{T, U}, A::MIOperator{T}, b::MIVector{T};
                  tol::Real = 1.0e-7, m::Int = min(Ks.maxiter, size(A, 1)), ishermitian::Bool = LinearAlgebra.ishermitian(A),
                  opnorm = nothing, iop::Int = 0, init::Int = 0, kw...) where {T, U}
    grow = m > Ks.maxiter
    check(ccall((:expv_mi_arnoldi, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ref{ArnoldiOpts}),
                handle(Ks), A.h, b.ptr, DEVICE, opts(m, tol, iop, init, ishermitian; flags = DEFER_TAIL)), ctx().h)
    grow && ((Ks.V, Ks.H) = views_of(T, U, handle(Ks), size(Ks.V, 1)); Ks.maxiter = m)         # resize!(Ks, m) happened inside (:355-357)
    sync_fields!(Ks)
end
# lanczos!(Ks, A, b; tol, m, ...)                                                              (src/arnoldi.jl:456-490)
function lanczos!(Ks::MIKs{T, U}, A::MIOperator{T}, b::MIVector{T};
                  tol::Real = 1.0e-7, m::Int = min(Ks.maxiter, size(A, 1)), opnorm = nothing, init::Int = 0, kw...) where {T, U}
    check(ccall((:expv_mi_lanczos, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ref{ArnoldiOpts}),
                handle(Ks), A.h, b.ptr, DEVICE, opts(m, tol, 0, init, 1; flags = DEFER_TAIL)), ctx().h)
    sync_fields!(Ks)
end
# arnoldi(A, b; m, ishermitian, kwargs...)                                                     (src/arnoldi.jl:161-180)
function arnoldi(A::MIOperator{T}, b::MIVector{T}; m = min(30, size(A, 1)),
                 ishermitian = LinearAlgebra.ishermitian(A), kw...) where {T}
    Ks = mi_subspace(T, ishermitian ? real(T) : T, length(b), m)
    arnoldi!(Ks, A, b; m = m, ishermitian = ishermitian, kw...)
end

# expv!(w, t, Ks; cache, expmethod)                                                           (src/krylov_phiv.jl:200-280)
# (the m x m exponential runs on the host inside the library: north_star; `cache` / `expmethod` are accepted and unused)
# Two methods with the reference's own split of `t` (krylov_phiv.jl:200-203 `t::Real`, :252-255 `t::Complex` with a complex w): one
# method on `t::Number` would be MORE specific than the reference's in w and Ks and LESS specific in t -- ambiguous for every call.
function _expv_ks!(w::MIVector{Tw}, t::Number, Ks) where {Tw}
    check(ccall((:expv_mi_expv_ks, lib), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Ptr{Cvoid}, Cint, Cint),
                handle(Ks), real(t), imag(t), w.ptr, DEVICE, dtype(Tw)), ctx().h)
    w
end
expv!(w::MIVector{Tw}, t::Real, Ks::MIKs{T, U}; cache = nothing, expmethod = nothing) where {Tw, T, U} = _expv_ks!(w, t, Ks)
expv!(w::MIVector{Complex{Tw}}, t::Complex{Tt}, Ks::MIKs{T, U}; cache = nothing, expmethod = nothing) where {Tw, Tt, T, U} =
    _expv_ks!(w, t, Ks)
# expv(t, A, b; kwargs...) in ONE library call (workspace reuse, no v_{m+1})                   (src/krylov_phiv.jl:135-144)
function ExponentialUtilities._expv_hb(t::Tt, A::MIOperator{T}, b::MIVector{T}; expmethod = nothing, cache = nothing,
                                       m = min(30, size(A, 1)), tol = 1.0e-7, iop = 0,
                                       ishermitian = LinearAlgebra.ishermitian(A), opnorm = nothing) where {Tt, T}
    w = similar(b, promote_type(Tt, T), (length(b),))
    st = Ref(ExpvStats(0, 0, 0, 0, 0.0))
    check(ccall((:expv_mi_expv, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Cint, Ref{ArnoldiOpts}, Ref{ExpvStats}),
                ctx().h, A.h, real(t), imag(t), b.ptr, DEVICE, w.ptr, DEVICE, dtype(eltype(w)), opts(m, tol, iop, 0, ishermitian), st), ctx().h)
    w
end
# expv(t, A, b; mode = :error_estimate, ...)                                                  (src/krylov_phiv.jl:145-160)
# The reference's _expv_ee builds a HOST KrylovSubspace{T, U}(n, m), which cannot hold device vectors: the subspace is created on
# the device instead.  T = promote_type(typeof(t), eltype(A), eltype(b)) as there: a complex t (the Schroedinger case, -im) on a
# real operator needs the operator and b in the complex type -- MIOperator(ComplexF64.(A)) -- which the reference's generic mul!
# does implicitly.
function ExponentialUtilities._expv_ee(t::Tt, A::MIOperator{T}, b::MIVector{T}; m = min(30, size(A, 1)), tol = 1.0e-7, rtol = √(tol),
                                       ishermitian::Bool = LinearAlgebra.ishermitian(A), expmethod = nothing) where {Tt, T}
    Tp = promote_type(Tt, T)
    Tp == T || throw(ArgumentError("expv(...; mode = :error_estimate) with a $(Tt) time works in $(Tp): build the operator and b " *
                                   "in that type (MIOperator($(Tp).(A)), MIArray($(Tp).(b)))"))
    Ks = mi_subspace(T, ishermitian ? real(T) : T, length(b), m)
    w = similar(b, T, (length(b),))
    expv!(w, t, A, b, Ks, get_subspace_cache(Ks); atol = tol, rtol = rtol, ishermitian = ishermitian)
end
# phiv!(w, t, Ks, k; cache, correct, errest, expmethod) / _phiv!                              (src/krylov_phiv.jl:607-653)
function _phiv!(w::MIMatrix{Tw}, t::Number, Ks::MIKs{T, U}, k::Integer, cache, correct, expmethod) where {Tw, T, U}
    err = Ref{Cdouble}(0)
    check(ccall((:expv_mi_phiv_ks, lib), Cint, (Ptr{Cvoid}, Cdouble, Cdouble, Cint, Cint, Ptr{Cvoid}, Int64, Cint, Cint, Ref{Cdouble}),
                handle(Ks), real(t), imag(t), k, correct, w.ptr, ld(w), DEVICE, dtype(Tw), err), ctx().h)
    w, err[]
end
function phiv!(w::MIMatrix, t::Number, Ks::MIKs, k::Integer; cache = nothing, correct = false, errest = false, expmethod = nothing)
    w, err = _phiv!(w, t, Ks, k, cache, correct, expmethod)
    errest ? (w, err) : w
end
function phiv(t, A::MIOperator{T}, b::MIVector{T}, k; cache = nothing, correct = false, errest = false, kwargs_arnoldi...) where {T}
    Ks = arnoldi(A, b; kwargs_arnoldi...)
    w = MIArray{promote_type(typeof(t), T)}(undef, length(b), k + 1)
    phiv!(w, t, Ks, k; cache = cache, correct = correct, errest = errest)
end
function phiv(t, Ks::MIKs{T, U}, k; kwargs...) where {T, U}
    w = MIArray{promote_type(typeof(t), T)}(undef, HANDLES[Ks].rows, k + 1)
    phiv!(w, t, Ks, k; kwargs...)
end
function expv(t::Tt, Ks::MIKs{T, U}; expmethod = nothing, kwargs...) where {Tt, T, U}                  # krylov_phiv.jl:161-168
    w = MIArray{promote_type(Tt, T)}(undef, HANDLES[Ks].rows)
    expv!(w, t, Ks; kwargs...)
end

# ---- error-estimate mode (Hermitian only)                                          (src/krylov_phiv_error_estimate.jl) ----
struct MISubspaceCache end                                                     # stands in for StegrCache: the library owns the scratch
get_subspace_cache(Ks::MIKs{T, U}) where {T, U <: Real} = MISubspaceCache()
get_subspace_cache(Ks::MIKs{T, U}) where {T, U <: Complex} =
    error("Subspace exponential caches not yet available for non-Hermitian matrices.")                 # :97
function expv!(w::MIVector{T}, t::Number, A::MIOperator{T}, b::MIVector{T}, Ks::MIKs{T, B}, cache::MISubspaceCache;
               atol::Real = 1.0e-8, rtol::Real = 1.0e-4, m = min(Ks.maxiter, size(A, 1)),
               ishermitian::Bool = LinearAlgebra.ishermitian(A), verbose::Bool = false, expmethod = nothing) where {T, B}
    check(ccall((:expv_mi_expv_error_estimate, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Cdouble, Cdouble, Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Cdouble, Cdouble, Cint, Cint),
                handle(Ks), A.h, real(t), imag(t), b.ptr, DEVICE, w.ptr, DEVICE, atol, rtol, m, ishermitian), ctx().h)
    sync_fields!(Ks)
    w
end

# ---- time stepping                                                                    (src/krylov_phiv_adaptive.jl) ----
# _phiv_timestep_caches(u_prototype, maxiter, p)  (:502-511): the tuple shape (u, W, P, Ks, phiv_cache) is an
# implementation detail of the reference; here ONE opaque handle plays that role.
mutable struct MITimestepCaches
    h::Ptr{Cvoid}
    ts1::Vector{Float64}                                                          # _singleton_ts slot (:237-242)
end
function _phiv_timestep_caches(u_prototype::MIVector{T}, maxiter::Int, p::Int) where {T}
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:expv_mi_timestep_caches_create, lib), Cint, (Ptr{Cvoid}, Cint, Int64, Cint, Cint, Ref{Ptr{Cvoid}}),
                ctx().h, dtype(T), length(u_prototype), maxiter, p, r), ctx().h)
    c = MITimestepCaches(r[], [0.0])
    finalizer(c -> ccall((:expv_mi_timestep_caches_destroy, lib), Cint, (Ptr{Cvoid},), c.h), c)
    c
end
const PRINTLN = Ref{Ptr{Cvoid}}(C_NULL)
_println_cb(line::Cstring, ::Ptr{Cvoid}) = (println(unsafe_string(line)); nothing)
println_ptr() = (PRINTLN[] == C_NULL && (PRINTLN[] = @cfunction(_println_cb, Cvoid, (Cstring, Ptr{Cvoid}))); PRINTLN[])
# the library's slow-progress notice (>= 10^4 accepted sub-steps without step growth: the reference's controller keeps a tiny seed
# step, krylov_phiv_adaptive.jl:391-417) reaches the user as a warning even without `verbose`
const WARNLN = Ref{Ptr{Cvoid}}(C_NULL)
_warn_cb(line::Cstring, ::Ptr{Cvoid}) = (@warn unsafe_string(line); nothing)
warn_ptr() = (WARNLN[] == C_NULL && (WARNLN[] = @cfunction(_warn_cb, Cvoid, (Cstring, Ptr{Cvoid}))); WARNLN[])

# phiv_timestep!(U, ts, A, B; ...)  -- the whole controller runs in the library                 (:260-453)
function phiv_timestep!(U::MIVecOrMat{T}, ts::AbstractVector{tType}, A::MIOperator{T}, B::MIVecOrMat{T};
                        tau::Real = 0.0, m::Int = min(10, size(A, 1)), tol::Real = 1.0e-7, opnorm = nothing, iop::Int = 0,
                        correct::Bool = false, caches = nothing, adaptive = false, delta::Real = 1.2,
                        ishermitian::Bool = LinearAlgebra.ishermitian(A), gamma::Real = 0.8, NA::Int = 0,
                        verbose = false) where {T, tType <: Real}
    length(ts) == ncols(U) || throw(AssertionError("Dimension mismatch"))                              # :307
    (size(U, 1) == size(A, 1) == size(B, 1)) || throw(AssertionError("Dimension mismatch"))            # :308
    tsv = ts isa Vector{Float64} ? ts : Vector{Float64}(ts)                     # sorted IN PLACE by the library, like sort!(ts) (:297)
    has_opn, opn = 0, 0.0
    if opnorm !== nothing
        has_opn, opn = 1, Float64(opnorm isa Number ? opnorm : opnorm(A, Inf))  # a number or a function (:276-281)
    end
    o = Ref(TimestepOpts(tau, tol, delta, gamma, opn, has_opn, m, iop, correct, adaptive, ishermitian, verbose, 0, 0, 0, NA,
                         verbose ? println_ptr() : warn_ptr(), C_NULL))
    st = Ref(TimestepStats(0, 0, 0, 0, 0, 0))
    check(ccall((:expv_mi_phiv_timestep, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cvoid}, Int64, Cint, Cint, Ptr{Cvoid}, Int64, Cint,
                 Ref{TimestepOpts}, Ptr{Cvoid}, Ref{TimestepStats}),
                ctx().h, A.h, length(tsv), tsv, B.ptr, ld(B), ncols(B), DEVICE, U.ptr, ld(U), DEVICE, o,
                caches === nothing ? C_NULL : caches.h, st), ctx().h)
    tsv === ts || copyto!(ts, tsv)
    U
end
# scalar-time forms (:184-232, :99-114) and the allocating front ends (:57-67)
_singleton_ts(caches::MITimestepCaches, t::Float64) = (caches.ts1[1] = t; caches.ts1)
_singleton_ts(_, t) = [Float64(t)]
function phiv_timestep!(u::MIVector{T}, t::Real, A::MIOperator{T}, B::MIMatrix{T}; caches = nothing, kwargs...) where {T}
    phiv_timestep!(u, _singleton_ts(caches, t), A, B; caches = caches, kwargs...)
    u
end
function expv_timestep!(u::MIVector{T}, t::Real, A::MIOperator{T}, b::MIVector{T}; caches = nothing, kwargs...) where {T}
    phiv_timestep!(u, _singleton_ts(caches, t), A, b; caches = caches, kwargs...)
    u
end
expv_timestep!(U::MIMatrix{T}, ts::AbstractVector{<:Real}, A::MIOperator{T}, b::MIVector{T}; kwargs...) where {T} =
    phiv_timestep!(U, ts, A, b; kwargs...)
phiv_timestep(ts::Vector{<:Real}, A::MIOperator{T}, B::MIVecOrMat{T}; kwargs...) where {T} =
    phiv_timestep!(MIArray{T}(undef, size(A, 1), length(ts)), ts, A, B; kwargs...)
phiv_timestep(t::Real, A::MIOperator{T}, B::MIVecOrMat{T}; kwargs...) where {T} =
    phiv_timestep!(MIArray{T}(undef, size(A, 1)), [Float64(t)], A, B; kwargs...)
expv_timestep(ts::Vector{<:Real}, A::MIOperator{T}, b::MIVector{T}; kwargs...) where {T} = phiv_timestep(ts, A, b; kwargs...)
expv_timestep(t::Real, A::MIOperator{T}, b::MIVector{T}; kwargs...) where {T} = phiv_timestep(t, A, b; kwargs...)

# ---- kiops(tau_out, A, u; ...)                                                                      (src/kiops.jl:57-281) ----
# Returns (w, stats) like the reference: w is n x size(tau_out, 2) (= n x 1: the only reachable case, see DESIGN.md),
# stats = (step, reject, krystep, exps, m).  The reference is real-only (kiops.jl:89, arnoldi.jl:197-200); a ComplexF64
# operator runs the documented mathematical extension.
function kiops(tau_out, A::MIOperator{T}, u::MIVecOrMat{T}; mmin::Int = 10, mmax::Int = 128, m::Int = min(mmin, mmax),
               tol::Real = 1.0e-7, opnorm = nothing, iop::Int = 2, ishermitian::Bool = LinearAlgebra.ishermitian(A),
               task1::Bool = false) where {T}
    taus = Float64.(vec(collect(tau_out)))                                    # linear indexing (kiops.jl:248)
    w = MIArray{T}(undef, size(A, 1), 1)
    o = Ref(KiopsOpts(mmin, mmax, m, iop, ishermitian, task1, 0, 0, tol))
    st = zeros(Int64, 5)
    check(ccall((:expv_mi_kiops, lib), Cint,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cdouble}, Cint, Cint, Ptr{Cvoid}, Int64, Cint, Cint, Ptr{Cvoid}, Int64, Cint,
                 Ref{KiopsOpts}, Ptr{Int64}),
                ctx().h, A.h, taus, length(taus), size(tau_out, 2), u.ptr, ld(u), ncols(u), DEVICE, w.ptr, ld(w), DEVICE, o, st), ctx().h)
    w, (st[1], st[2], st[3], st[4], st[5])
end

# ---- batch of independent problems over the GPUs of a node (BASELINE config 5; no reference counterpart: a host `for`) ----
# vals: nnz x nprob (values of problem p in column p, CSR order of `pattern`), B and the result: n x nprob host matrices.
function expv_batch(ts::Vector{Float64}, pattern::SparseMatrixCSC, vals::Matrix{T}, B::Matrix{T}; devices = [0],
                    m::Int = 30, tol::Real = 1.0e-7, iop::Int = 0, ishermitian::Bool = false) where {T <: MIScalar}
    P = SparseMatrixCSC(transpose(pattern))                                   # CSR of `pattern` = CSC of its transpose
    n, nprob = size(pattern, 1), size(B, 2)
    rowptr = Int32.(P.colptr .- 1)
    colind = Int32.(P.rowval .- 1)
    ctxs = [Ctx(d) for d in devices]
    W = Matrix{T}(undef, n, nprob)
    mused = zeros(Int32, nprob)
    hs = [c.h for c in ctxs]
    code = ccall((:expv_mi_expv_batch_multi, lib), Cint,
                 (Ptr{Ptr{Cvoid}}, Cint, Cint, Int64, Cint, Ptr{Int32}, Ptr{Int32}, Ptr{T}, Int64, Ptr{Cdouble}, Ptr{T}, Int64,
                  Ptr{T}, Int64, Cint, Ref{ArnoldiOpts}, Ptr{Int32}),
                 hs, length(hs), dtype(T), n, nprob, rowptr, colind, vals, size(vals, 1), ts, B, n, W, n, HOST,
                 opts(m, tol, iop, 0, ishermitian), mused)
    code == 0 || foreach(c -> check(code, c.h), ctxs)
    W, mused
end

# ---- the same batch with ONE Julia process per GPU: the final gather over RCCL (north_star: "RCCL over xGMI for the final gather only") ----
# Each process solves its contiguous block with `expv_batch(...; devices = [local_gpu])`-style calls or its own loop, leaves the block in an
# MIArray, and `allgather_columns!` brings the n x nprob result to every rank: one ncclAllGather on the context's stream.  The 128-byte id comes
# from rank 0 (`rccl_unique_id()`) and reaches the other ranks by the host's own means (MPI.jl bcast, Distributed.jl, a file).
rccl_available() = ccall((:expv_mi_rccl_available, lib), Cint, ()) == 1
function rccl_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:expv_mi_rccl_unique_id, lib), Cint, (Ptr{UInt8},), id), C_NULL)
    id
end
mutable struct RcclComm
    h::Ptr{Cvoid}
    nranks::Int
    rank::Int
    function RcclComm(id::Vector{UInt8}, nranks::Integer, rank::Integer)
        length(id) == 128 || throw(ArgumentError("unique id: 128 bytes"))
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:expv_mi_comm_create, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint, Ptr{Ptr{Cvoid}}), ctx().h, id, nranks, rank, h), ctx().h)
        c = new(h[], nranks, rank)
        finalizer(x -> ccall((:expv_mi_comm_destroy, lib), Cint, (Ptr{Cvoid},), x.h), c)
        c
    end
end
# recv[:, r * cols + 1 : (r + 1) * cols] = rank r's `send` (n x cols, contiguous): the blocks of `dist.shard_range` with equal shares
function allgather_columns!(recv::MIArray{T}, send::MIArray{T}, comm::RcclComm) where {T}
    length(recv) == comm.nranks * length(send) || throw(DimensionMismatch("recv must hold nranks blocks of send's size"))
    check(ccall((:expv_mi_gather_rccl, lib), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int64, Cint),
                comm.h, send.ptr, recv.ptr, length(send), dtype(T)), ctx().h)
    recv
end

end # module
