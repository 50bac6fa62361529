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

const libqgd = joinpath(@__DIR__, "..", "quantumgatedesign.jl_amd", "csrc", "libqgd_hip.so")

struct ProblemDesc            # qgd_problem_desc, include/qgd.h
    N::Int32; n_cols::Int32; n_ops::Int32; n_ess::Int32; order::Int32; nsteps::Int32
    tf::Float64
    system_sym::Ptr{Float64}; system_asym::Ptr{Float64}
    sym_ops::Ptr{Float64}; asym_ops::Ptr{Float64}
    u0::Ptr{Float64}; v0::Ptr{Float64}; guard::Ptr{Float64}
    device::Int32; reserved::Int32
end

mutable struct DeviceProblem
    handle::Ptr{Cvoid}
    order::Int
    nsteps::Int
    basis_key::UInt
end

function check(h, rc)
    rc == 0 && return
    msg = unsafe_string(ccall((:qgd_last_error, libqgd), Cstring, (Ptr{Cvoid},), h))
    rc == 1 ? throw(ArgumentError(msg)) : error("qgd error $rc: $msg")
end

"One handle per (prob, order); dense column-major copies are made for the call only."
function DeviceProblem(prob::SchrodingerProb, order::Integer; device::Integer=0)
    N = prob.N_tot_levels
    ssym, sasym = Matrix{Float64}(prob.system_sym), Matrix{Float64}(prob.system_asym)
    sym = reduce(hcat, [vec(Matrix{Float64}(op)) for op in prob.sym_operators]; init=zeros(N*N, 0))
    asym = reduce(hcat, [vec(Matrix{Float64}(op)) for op in prob.asym_operators]; init=zeros(N*N, 0))
    u0, v0 = Matrix{Float64}(reshape(prob.u0, N, :)), Matrix{Float64}(reshape(prob.v0, N, :))
    W = Matrix{Float64}(prob.guard_subspace_projector)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve ssym sasym sym asym u0 v0 W begin
        d = ProblemDesc(N, size(u0, 2), prob.N_operators, prob.N_ess_levels, order, prob.nsteps, prob.tf,
                        pointer(ssym), pointer(sasym), pointer(sym), pointer(asym),
                        pointer(u0), pointer(v0), pointer(W), device, 0)
        rc = ccall((:qgd_create, libqgd), Cint, (Ref{ProblemDesc}, Ref{Ptr{Cvoid}}), d, h)
    end
    check(C_NULL, rc)
    dp = DeviceProblem(h[], order, prob.nsteps, UInt(0))
    finalizer(x -> ccall((:qgd_destroy, libqgd), Cvoid, (Ptr{Cvoid},), x.handle), dp)
    return dp
end

"Control basis G[n,d,l] = d/dpcof_l (p^(d)(t_n)/d!) from the package's own eval_grad_*_derivative!
(all in-scope controls are linear in pcof).  C-ordered [nt][m+1][N_coeff] == Julia Array (N_coeff, m+1, nt)."
function set_controls!(dp::DeviceProblem, prob, controls, pcof)
    key = hash((objectid(controls), prob.nsteps, prob.tf))
    key == dp.basis_key && return
    m, nt, dt = div(dp.order, 2), prob.nsteps + 1, prob.tf / prob.nsteps
    Gp, Gq, ncoef = Vector{Array{Float64,3}}(), Vector{Array{Float64,3}}(), Int32[]
    for k in 1:prob.N_operators
        c = controls[k]; nc = c.N_coeff; push!(ncoef, nc)
        gp, gq = zeros(nc, m + 1, nt), zeros(nc, m + 1, nt)
        lp = QuantumGateDesign.get_control_vector_slice(pcof, controls, k)
        for n in 0:nt-1, d in 0:m
            eval_grad_p_derivative!(view(gp, :, 1 + d, 1 + n), c, n * dt, lp, d)
            eval_grad_q_derivative!(view(gq, :, 1 + d, 1 + n), c, n * dt, lp, d)
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
end

const _cache = IdDict{Any,DeviceProblem}()
function device_problem(prob, order)
    dp = get!(() -> DeviceProblem(prob, order), _cache, (prob, order))
    if dp.nsteps != prob.nsteps      # scripts mutate prob.nsteps (examples/cnot3_optimize_gate.jl:51-52)
        check(dp.handle, ccall((:qgd_set_nsteps, libqgd), Cint, (Ptr{Cvoid}, Int32, Float64), dp.handle, prob.nsteps, prob.tf))
        dp.nsteps = prob.nsteps; dp.basis_key = UInt(0)
    end
    return dp
end

"Drop-in for eval_forward!(uv_history, prob, controls, pcof; order) -- src/forward_evolution.jl:33-70."
function hip_eval_forward!(uv_history::Array{Float64,4}, prob::SchrodingerProb, controls, pcof::Vector{Float64}; order::Int=2)
    dp = device_problem(prob, order)
    set_controls!(dp, prob, controls, pcof)
    out3 = zeros(3)
    check(dp.handle, ccall((:qgd_eval_forward, libqgd), Cint,
          (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}), dp.handle, pcof, length(pcof), uv_history, out3))
    return nothing
end

"Drop-in for discrete_adjoint!(grad, history, lambda_history, adjoint_forcing, prob, controls, pcof, target;
order, history_precomputed) -- src/eval_grad_discrete_adjoint.jl:107-160."
function hip_discrete_adjoint!(grad::Vector{Float64}, history::Array{Float64,4}, lambda_history::Array{Float64,4},
        adjoint_forcing::Array{Float64,3}, prob::SchrodingerProb, controls, pcof::Vector{Float64},
        target::AbstractMatrix{<:Number}; order::Int=2, history_precomputed::Bool=false)
    dp = device_problem(prob, order)
    set_controls!(dp, prob, controls, pcof)
    tr = Matrix{Float64}(vcat(real(target), imag(target)))       # as eval_grad_discrete_adjoint.jl:126
    check(dp.handle, ccall((:qgd_set_target, libqgd), Cint, (Ptr{Cvoid}, Ptr{Float64}), dp.handle, tr))
    out3 = zeros(3)
    check(dp.handle, ccall((:qgd_discrete_adjoint, libqgd), Cint,
          (Ptr{Cvoid}, Ptr{Float64}, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
          dp.handle, pcof, length(pcof), history_precomputed, grad, history, lambda_history, adjoint_forcing, out3))
    return grad
end

"""
    hip_eval_grad_forced(prob, controls, pcof, target; order=2)

Drop-in for `eval_grad_forced` (src/eval_grad_forced.jl:17-60): the gradient by forward
sensitivities, all control parameters in one device call.
"""
function hip_eval_grad_forced(prob::SchrodingerProb, controls, pcof::Vector{Float64},
        target::AbstractMatrix{<:Number}; order::Int=2)
    dp = device_problem(prob, order)
    set_controls!(dp, prob, controls, pcof)
    tr = Matrix{Float64}(vcat(real(target), imag(target)))
    check(dp.handle, ccall((:qgd_set_target, libqgd), Cint, (Ptr{Cvoid}, Ptr{Float64}), dp.handle, tr))
    grad = zeros(length(pcof))
    check(dp.handle, ccall((:qgd_eval_grad_forced, libqgd), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}),
          dp.handle, pcof, length(pcof), grad))
    return grad
end

end # module
