# QuantumGateDesignHIP.jl -- the reference-side binding a maintainer of
# QuantumGateDesign.jl would add to route eval_forward!/discrete_adjoint! through
# libqgd_hip.so (include/qgd.h).  It follows the package's one existing FFI precedent, the
# Fortran ccall of src/Controls/FortranBSpline.jl:257-265 (library path next to the module,
# Ref/pointer arguments).  NOT executed in this repository's CI: Julia is not in the image.
# The Python mirror in quantumgatedesign.jl_amd/ exercises exactly these entry points.
module QuantumGateDesignHIP

using QuantumGateDesign
import QuantumGateDesign: SchrodingerProb, eval_forward!, discrete_adjoint!,
                          get_number_of_control_parameters, eval_grad_p_derivative!,
                          eval_grad_q_derivative!
using SparseArrays: SparseMatrixCSC, sparse

const libqgd = joinpath(@__DIR__, "..", "quantumgatedesign.jl_amd", "csrc", "libqgd_hip.so")

struct ProblemDesc            # qgd_problem_desc, include/qgd.h
    N::Int32; n_cols::Int32; n_ops::Int32; n_ess::Int32; order::Int32; nsteps::Int32
    tf::Float64
    system_sym::Ptr{Float64}; system_asym::Ptr{Float64}
    sym_ops::Ptr{Float64}; asym_ops::Ptr{Float64}
    u0::Ptr{Float64}; v0::Ptr{Float64}; guard::Ptr{Float64}
    device::Int32; reserved::Int32
end

struct CSC                    # qgd_csc, include/qgd.h: the three arrays of a SparseMatrixCSC{Float64,Int64}
    colptr::Ptr{Int64}; rowval::Ptr{Int64}; nzval::Ptr{Float64}
    index_base::Int32; reserved::Int32
end

mutable struct DeviceProblem
    handle::Ptr{Cvoid}
    order::Int
    nsteps::Int
    basis_key::UInt
    pinned::Vector{Any}      # output arrays registered with the handle (at most MAX_PINNED, oldest unregistered first)
    general::Bool            # controls that are not linear in pcof: tables + Jacobian per evaluation, NULL pcof
    general_pcof::Vector{Float64}
    cost_type::Symbol        # what the handle is set to (ccall only on change)
    lambda_derivatives::Bool
    last_out3::Vector{Float64}   # the three scalars of the last evaluation (objective_terms / last_objective)
end

function check(h, rc)
    rc == 0 && return
    msg = unsafe_string(ccall((:qgd_last_error, libqgd), Cstring, (Ptr{Cvoid},), h))
    rc == 1 ? throw(ArgumentError(msg)) : error("qgd error $rc: $msg")
end

"One handle per (prob, order).  Operators stored as SparseMatrixCSC (DispersiveProblem's default, sparse_rep=true)
go to the library as they are (qgd_create_csc: colptr/rowval/nzval, 1-based); dense ones as column-major copies made
for the call only."
function DeviceProblem(prob::SchrodingerProb, order::Integer; device::Integer=0, defer_grid::Bool=false)
    N = prob.N_tot_levels
    u0, v0 = Matrix{Float64}(reshape(prob.u0, N, :)), Matrix{Float64}(reshape(prob.v0, N, :))
    W = Matrix{Float64}(prob.guard_subspace_projector)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    if prob.system_sym isa SparseMatrixCSC
        mats = SparseMatrixCSC{Float64,Int64}[prob.system_sym, prob.system_asym, prob.sym_operators..., prob.asym_operators...]
        GC.@preserve mats u0 v0 W begin
            csc = [CSC(pointer(a.colptr), pointer(a.rowval), pointer(a.nzval), 1, 0) for a in mats]
            nop = prob.N_operators
            d = ProblemDesc(N, size(u0, 2), nop, prob.N_ess_levels, order, prob.nsteps, prob.tf,
                            C_NULL, C_NULL, C_NULL, C_NULL, pointer(u0), pointer(v0), pointer(W), device, defer_grid ? 1 : 0)
            rc = ccall((:qgd_create_csc, libqgd), Cint,
                       (Ref{ProblemDesc}, Ref{CSC}, Ref{CSC}, Ptr{CSC}, Ptr{CSC}, Ref{Ptr{Cvoid}}),
                       d, csc[1], csc[2], pointer(csc, 3), pointer(csc, 3 + nop), h)
        end
    else
        ssym, sasym = Matrix{Float64}(prob.system_sym), Matrix{Float64}(prob.system_asym)
        sym = reduce(hcat, [vec(Matrix{Float64}(op)) for op in prob.sym_operators]; init=zeros(N*N, 0))
        asym = reduce(hcat, [vec(Matrix{Float64}(op)) for op in prob.asym_operators]; init=zeros(N*N, 0))
        GC.@preserve ssym sasym sym asym u0 v0 W begin
            d = ProblemDesc(N, size(u0, 2), prob.N_operators, prob.N_ess_levels, order, prob.nsteps, prob.tf,
                            pointer(ssym), pointer(sasym), pointer(sym), pointer(asym),
                            pointer(u0), pointer(v0), pointer(W), device, defer_grid ? 1 : 0)
            rc = ccall((:qgd_create, libqgd), Cint, (Ref{ProblemDesc}, Ref{Ptr{Cvoid}}), d, h)
        end
    end
    check(C_NULL, rc)
    dp = DeviceProblem(h[], order, prob.nsteps, UInt(0), Any[], false, Float64[], :Infidelity, false, zeros(3))
    finalizer(x -> ccall((:qgd_destroy, libqgd), Cvoid, (Ptr{Cvoid},), x.handle), dp)
    return dp
end

"Pin an output array that is handed to hip_discrete_adjoint! on every iteration (optimize_gate allocates
state_history, lambda_history and adjoint_forcing once, src/ipopt_optimal_control.jl:207-214): its download then
runs at PCIe speed beside the adjoint sweep.  The array is kept alive by the handle."
const MAX_PINNED = 6        # two sets of (history, lambda_history, adjoint_forcing): a caller that passes fresh arrays
                            # on every call must not grow pinned memory without bound
function pin!(dp::DeviceProblem, a::Array{Float64})
    any(x -> x === a, dp.pinned) && return a
    while length(dp.pinned) >= MAX_PINNED
        old = popfirst!(dp.pinned)
        ccall((:qgd_unregister_host_buffer, libqgd), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), dp.handle, old)
    end
    check(dp.handle, ccall((:qgd_register_host_buffer, libqgd), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
          dp.handle, a, sizeof(a)))
    push!(dp.pinned, a)
    return a
end

"Control basis G[n,d,l] = d/dpcof_l (p^(d)(t_n)/d!) from the package's own eval_grad_*_derivative!.
Every family the package ships is LINEAR in pcof (B-splines with or without carriers, GRAPE, Hermite): the basis is
uploaded once and an evaluation ships only pcof.  AbstractControl is an open protocol (src/Controls/Control.jl:6-27),
so linearity is probed -- the Jacobian must not depend on pcof.  A control that fails the probe takes the GENERAL path
of include/qgd.h: at every new pcof the tables (fill_p_mat!/fill_q_mat! over the grid, qgd_set_control_tables) and their
Jacobian at that pcof (qgd_set_control_basis) are uploaded, and the evaluation is called with a NULL pcof.
Returns the (pointer, length) to pass as pcof."
function set_controls!(dp::DeviceProblem, prob, controls, pcof)
    key = hash((objectid(controls), prob.nsteps, prob.tf))
    if key == dp.basis_key
        dp.general || return (pointer(pcof), length(pcof))
        dp.general_pcof == pcof && return (Ptr{Float64}(C_NULL), 0)
    end
    n_lo, n_hi = time_window(dp)           # the whole grid unless comm_init!(...; shard=:time) gave this rank a window
    m, nt, dt = div(dp.order, 2), n_hi - n_lo + 1, prob.tf / prob.nsteps
    Gp, Gq, ncoef = Vector{Array{Float64,3}}(), Vector{Array{Float64,3}}(), Int32[]
    linear = true
    for k in 1:prob.N_operators
        c = controls[k]; nc = c.N_coeff; push!(ncoef, nc)
        gp, gq = zeros(nc, m + 1, nt), zeros(nc, m + 1, nt)
        lp = Vector{Float64}(QuantumGateDesign.get_control_vector_slice(pcof, controls, k))
        probe, g1, g2 = lp .+ 0.37 .* (1 .+ abs.(lp)), zeros(nc), zeros(nc)
        for n in (0, div(nt, 3), nt - 1), d in 0:m            # linearity probe at three time points
            for (f!) in (eval_grad_p_derivative!, eval_grad_q_derivative!)
                f!(g1, c, (n_lo + n) * dt, lp, d); f!(g2, c, (n_lo + n) * dt, probe, d)
                linear &= maximum(abs.(g1 .- g2)) <= 1e-12 * max(1.0, maximum(abs.(g1)))
            end
        end
        for n in 0:nt-1, d in 0:m
            eval_grad_p_derivative!(view(gp, :, 1 + d, 1 + n), c, (n_lo + n) * dt, lp, d)
            eval_grad_q_derivative!(view(gq, :, 1 + d, 1 + n), c, (n_lo + n) * dt, lp, d)
            gp[:, 1 + d, 1 + n] ./= factorial(d); gq[:, 1 + d, 1 + n] ./= factorial(d)
        end
        push!(Gp, gp); push!(Gq, gq)
    end
    GC.@preserve Gp Gq begin
        pp, pq = [pointer(g) for g in Gp], [pointer(g) for g in Gq]
        check(dp.handle, ccall((:qgd_set_control_basis, libqgd), Cint,
              (Ptr{Cvoid}, Ptr{Int32}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}), dp.handle, ncoef, pp, pq))
    end
    dp.basis_key = key
    dp.general = !linear
    linear && return (pointer(pcof), length(pcof))
    # general path: the tables themselves, Julia layout [(1+m), N_operators, nsteps+1] (fill_p_mat! stacked over the grid)
    pt, qt = zeros(m + 1, prob.N_operators, nt), zeros(m + 1, prob.N_operators, nt)
    for n in 0:nt-1
        QuantumGateDesign.fill_p_mat!(view(pt, :, :, 1 + n), controls, (n_lo + n) * dt, pcof)
        QuantumGateDesign.fill_q_mat!(view(qt, :, :, 1 + n), controls, (n_lo + n) * dt, pcof)
    end
    check(dp.handle, ccall((:qgd_set_control_tables, libqgd), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), dp.handle, pt, qt))
    dp.general_pcof = copy(pcof)
    return (Ptr{Float64}(C_NULL), 0)
end

# Handle cache: at most MAX_HANDLES (a cnot3 handle holds ~300 MB of device memory), least recently used first out;
# keyed by the prob OBJECT, so a copy(prob) gets its own handle and release!(prob) frees it explicitly.
const MAX_HANDLES = 4
const _cache = Vector{Tuple{Any,Int,DeviceProblem}}()
function device_problem(prob, order)
    i = findfirst(e -> e[1] === prob && e[2] == order, _cache)
    if i === nothing
        length(_cache) >= MAX_HANDLES && finalize(popfirst!(_cache)[3])
        push!(_cache, (prob, order, DeviceProblem(prob, order)))
    else
        push!(_cache, splice!(_cache, i))
    end
    dp = _cache[end][3]
    if dp.nsteps != prob.nsteps      # scripts mutate prob.nsteps (examples/cnot3_optimize_gate.jl:51-52)
        check(dp.handle, ccall((:qgd_set_nsteps, libqgd), Cint, (Ptr{Cvoid}, Int32, Float64), dp.handle, prob.nsteps, prob.tf))
        dp.nsteps = prob.nsteps; dp.basis_key = UInt(0)
    end
    return dp
end
release!(prob) = foreach(e -> finalize(e[3]), splice!(_cache, findall(e -> e[1] === prob, _cache)))

"infidelity + guard penalty of the last evaluation from the three scalars the device returns
(infidelity_real, guard_penalty_real: src/infidelity.jl:7-18, :56-96) -- no state history needed on the host."
objective_terms(out3, N_ess) = (1 - (out3[1]^2 + out3[2]^2) / N_ess^2, out3[3])

"Drop-in for eval_forward!(uv_history, prob, controls, pcof; order, saveEveryNsteps) -- src/forward_evolution.jl:33-70.
With saveEveryNsteps = s the array is [2N, 1+order/2, 1+div(nsteps,s), N_initial_conditions] and the library's re-layout
kernel fills it with every s-th time point (qgd_set_save_every; :104, :239-241)."
function hip_eval_forward!(uv_history::Array{Float64,4}, prob::SchrodingerProb, controls, pcof::Vector{Float64};
                           order::Int=2, saveEveryNsteps::Int=1)
    dp = device_problem(prob, order)
    set_cost_type!(dp, :Infidelity)
    pc_ptr, pc_len = set_controls!(dp, prob, controls, pcof)
    n_lo, n_hi = time_window(dp)           # (a time-partitioned handle fills the slots of its own window)
    size(uv_history, 3) == 1 + div(n_hi - n_lo, saveEveryNsteps) || throw(DimensionMismatch("uv_history: $(size(uv_history, 3)) time slots for saveEveryNsteps=$saveEveryNsteps over time points $n_lo:$n_hi"))
    check(dp.handle, ccall((:qgd_set_save_every, libqgd), Cint, (Ptr{Cvoid}, Int32), dp.handle, saveEveryNsteps))
    try
        GC.@preserve pcof check(dp.handle, ccall((:qgd_eval_forward, libqgd), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}), dp.handle, pc_ptr, pc_len, uv_history, dp.last_out3))
    finally
        ccall((:qgd_set_save_every, libqgd), Cint, (Ptr{Cvoid}, Int32), dp.handle, 1)
    end
    return nothing      # (as the reference; the three scalars: last_objective(prob, order))
end

"(infidelity, guard penalty) of the last evaluation of (prob, order) -- with :Tracking / :Norm: (cost, guard penalty)."
function last_objective(prob::SchrodingerProb, order::Integer)
    dp = device_problem(prob, order)
    return dp.cost_type == :Infidelity ? objective_terms(dp.last_out3, prob.N_ess_levels) : (dp.last_out3[1], dp.last_out3[3])
end

"The objective without the history: what optimize_gate's eval_f needs (INTEGRATION.md section 2, variant B)."
function hip_objective(prob::SchrodingerProb, controls, pcof::Vector{Float64}, target; order::Int=2)
    dp = device_problem(prob, order)
    set_cost_type!(dp, :Infidelity)         # (every entry point states the cost type it computes with)
    pc_ptr, pc_len = set_controls!(dp, prob, controls, pcof)
    tr = Matrix{Float64}(vcat(real(target), imag(target)))
    check(dp.handle, ccall((:qgd_set_target, libqgd), Cint, (Ptr{Cvoid}, Ptr{Float64}), dp.handle, tr))
    GC.@preserve pcof check(dp.handle, ccall((:qgd_eval_forward, libqgd), Cint,
          (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}), dp.handle, pc_ptr, pc_len, C_NULL, dp.last_out3))
    return objective_terms(dp.last_out3, prob.N_ess_levels)
end

"cost_type as in src/eval_grad_discrete_adjoint.jl:26-35; anything else throws like the reference."
function set_cost_type!(dp, cost_type)
    code = cost_type == :Infidelity ? 0 : cost_type == :Tracking ? 1 : cost_type == :Norm ? 2 : throw("Invalid cost type: $cost_type")
    dp.cost_type == cost_type && return
    check(dp.handle, ccall((:qgd_set_cost_type, libqgd), Cint, (Ptr{Cvoid}, Int32), dp.handle, code))
    dp.cost_type = cost_type
end

"Drop-in for discrete_adjoint!(grad, history, lambda_history, adjoint_forcing, prob, controls, pcof, target;
order, history_precomputed) -- src/eval_grad_discrete_adjoint.jl:107-160.  `lambda_derivatives=true` also fills
lambda_history[:, 2:end, :, :] the way eval_adjoint! leaves it (src/forward_evolution.jl:427-433); nothing in the package
reads those columns, so the default keeps them zero and the download small."
function hip_discrete_adjoint!(grad::Vector{Float64}, history::Union{Array{Float64,4},Nothing},
        lambda_history::Union{Array{Float64,4},Nothing}, adjoint_forcing::Union{Array{Float64,3},Nothing},
        prob::SchrodingerProb, controls, pcof::Vector{Float64},
        target::AbstractMatrix{<:Number}; order::Int=2, history_precomputed::Bool=false, cost_type=:Infidelity,
        lambda_derivatives::Bool=false)
    dp = device_problem(prob, order)
    pc_ptr, pc_len = set_controls!(dp, prob, controls, pcof)
    set_cost_type!(dp, cost_type)
    n_lo, n_hi = time_window(dp)          # (a time-partitioned handle returns its own window of time points)
    shape = (prob.real_system_size, 1 + div(order, 2), n_hi - n_lo + 1, prob.N_initial_conditions)
    for a in (history, lambda_history)
        a === nothing || size(a) == shape || throw(DimensionMismatch("history arrays must be $shape, got $(size(a))"))
    end
    adjoint_forcing === nothing || size(adjoint_forcing) == (shape[1], shape[3], shape[4]) || throw(DimensionMismatch("adjoint_forcing"))
    foreach(a -> a === nothing || pin!(dp, a), (history, lambda_history, adjoint_forcing))
    ptr(a) = a === nothing ? Ptr{Float64}(C_NULL) : pointer(a)
    tr = Matrix{Float64}(vcat(real(target), imag(target)))       # as eval_grad_discrete_adjoint.jl:126
    check(dp.handle, ccall((:qgd_set_target, libqgd), Cint, (Ptr{Cvoid}, Ptr{Float64}), dp.handle, tr))
    if dp.lambda_derivatives != lambda_derivatives      # (only on change: the setter makes the library re-zero a registered lambda_history)
        check(dp.handle, ccall((:qgd_set_lambda_derivatives, libqgd), Cint, (Ptr{Cvoid}, Int32), dp.handle, lambda_derivatives))
        dp.lambda_derivatives = lambda_derivatives
    end
    GC.@preserve pcof history lambda_history adjoint_forcing begin
        check(dp.handle, ccall((:qgd_discrete_adjoint, libqgd), Cint,
              (Ptr{Cvoid}, Ptr{Float64}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
              dp.handle, pc_ptr, pc_len, history_precomputed, grad, ptr(history), ptr(lambda_history), ptr(adjoint_forcing), dp.last_out3))
    end
    # returns the gradient, as the reference (src/eval_grad_discrete_adjoint.jl:158); the scalars of this pcof --
    # (infidelity, guard penalty), or (cost, guard penalty) with :Tracking / :Norm -- are last_objective(prob, order)
    return grad
end

"""
    hip_eval_grad_forced(prob, controls, pcof, target; order=2, cost_type=:Infidelity)

Drop-in for `eval_grad_forced` (src/eval_grad_forced.jl:17-60): the gradient by forward
sensitivities, all control parameters in one device call.
"""
function hip_eval_grad_forced(prob::SchrodingerProb, controls, pcof::Vector{Float64},
        target::AbstractMatrix{<:Number}; order::Int=2, cost_type=:Infidelity)
    dp = device_problem(prob, order)
    pc_ptr, pc_len = set_controls!(dp, prob, controls, pcof)
    set_cost_type!(dp, cost_type)
    tr = Matrix{Float64}(vcat(real(target), imag(target)))
    check(dp.handle, ccall((:qgd_set_target, libqgd), Cint, (Ptr{Cvoid}, Ptr{Float64}), dp.handle, tr))
    grad = zeros(length(pcof))
    GC.@preserve pcof check(dp.handle, ccall((:qgd_eval_grad_forced, libqgd), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}),
          dp.handle, pc_ptr, pc_len, grad))
    return grad
end

# ---- several GPUs: RCCL inside the library (include/qgd.h, "several GPUs behind ONE call") ------------------------------
# One Julia process (or task pinned to a thread) per GPU.  Rank 0 makes the id; the host carries its 128 bytes to the
# other ranks once (e.g. MPI.jl: `id = MPI.bcast(rank == 0 ? comm_unique_id() : nothing, 0, MPI.COMM_WORLD)`); after
# comm_init! the very same hip_discrete_adjoint! / hip_objective calls are collective -- every rank makes them with the
# same pcof, the library issues the all-gathers / all-reduce on its stream -- and every rank gets the full gradient.
# This replaces the reference's Threads.@threads loop over initial conditions (src/forward_evolution.jl:48,332).

"128-byte RCCL id (rank 0)."
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(C_NULL, ccall((:qgd_comm_unique_id, libqgd), Cint, (Ptr{UInt8},), id))
    return id
end

"Give the handle of (prob, order) on this process's GPU its communicator.  shard = :time (windows of the time grid per
rank; the controls' basis is then built for the rank's own window by the next set_controls!) or :columns (prob must hold
this rank's columns of u0, v0 -- and the caller passes its columns of the target -- with the global N_ess_levels)."
function comm_init!(prob::SchrodingerProb, order::Integer, id::Vector{UInt8}, rank::Integer, world::Integer; shard::Symbol=:time, device::Integer=rank)
    i = findfirst(e -> e[1] === prob && e[2] == order, _cache)
    # (QGD_CREATE_DEFER_GRID: a rank of a time partition never allocates the whole grid -- qgd_comm_init_rccl allocates its window)
    i === nothing && push!(_cache, (prob, order, DeviceProblem(prob, order; device=device, defer_grid=(shard == :time))))
    dp = device_problem(prob, order)
    check(dp.handle, ccall((:qgd_comm_init_rccl, libqgd), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32, Int32),
          dp.handle, id, rank, world, shard == :columns ? 1 : 0))
    dp.basis_key = UInt(0)          # (a time window re-allocates the grid: the basis is set again, for the window)
    return dp
end

"First and last global time point of this rank's window (qgd_get_partition); (0, nsteps) without a time partition.
A handle that works through a long grid in windows on ONE GPU (qgd_set_memory_budget, or more than 65 000 steps) owns
the whole grid as far as its caller is concerned -- control basis and output arrays cover every time point -- so only a
handle whose partition has more than one rank (out[7] = world) reports a window of its own."
function time_window(dp::DeviceProblem)
    out = zeros(Int32, 8)
    check(dp.handle, ccall((:qgd_get_partition, libqgd), Cint, (Ptr{Cvoid}, Ptr{Int32}), dp.handle, out))
    Int(out[7]) <= 1 && return 0, Int(out[8]) - 1
    return Int(out[1]), Int(out[2])
end

"Bound (milliseconds, default 30 000) on the host wait of a collective evaluation: past it the communicator is aborted
and the call throws with QGD_ERR_COMM instead of waiting for a rank that failed (qgd_set_comm_timeout)."
function comm_timeout!(dp::DeviceProblem, milliseconds::Real)
    check(dp.handle, ccall((:qgd_set_comm_timeout, libqgd), Cint, (Ptr{Cvoid}, Float64), dp.handle, milliseconds))
end

"Problems with N <= 4 levels, <= 4 initial conditions and <= 128 time points (the Rabi oscillator, the two-qubit CNOT) are
evaluated by a four-launch path of their own (qgd_set_small_path); `on = false` keeps this handle on the general kernels."
function small_path!(dp::DeviceProblem, on::Bool)
    check(dp.handle, ccall((:qgd_set_small_path, libqgd), Cint, (Ptr{Cvoid}, Int32), dp.handle, on ? 1 : 0))
end

end # module
