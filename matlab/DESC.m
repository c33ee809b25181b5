%% DESC -- drop-in replacement of the reference's Algorithms/DESC.m:14 (called at
%% Demo/compare_algorithms.m:72):   [R_est, R_init, S_vec] = DESC(Ind, RijMat, params)
%% PGD (DESC.m:16-261) -> GCW (:263) -> reweighted LAA refinement (:265-313), all on the MI355X; Ind / RijMat go to
%% the GPU once for the three stages (desc_amd_mex('desc', ...): desc_problem_upload).
function [R_est, R_init, S_vec] = DESC(Ind, RijMat, params)
    G = params.Gradient;
    if isa(G, 'HybridGradient') && G.strategy == 0
        % Adam keeps per-cycle state in the handle object: stage by stage
        S_vec = DESC_PGD(Ind, RijMat, params);
        [IndS, perm] = sortrows(double(Ind), [1 2]);
        R_init = desc_amd_mex('gcw', int32(IndS - 1), double(RijMat(:,:,perm)), S_vec(perm));
        disp('Rotation Initialized!')                 % DESC.m:283
        disp('Start DESC refinement ...')             % DESC.m:284
        R_est = desc_amd_mex('refine', int32(IndS - 1), double(RijMat(:,:,perm)), S_vec(perm), R_init);
        disp('DONE!')                                 % DESC.m:313
        return
    end
    if isfield(params, 'make_plots') && params.make_plots
        error('desc_amd:make_plots', 'params.make_plots=true is served by the Python host layer only (desc_amd.DESC); set make_plots=false.');
    end
    [IndS, perm] = sortrows(double(Ind), [1 2]);
    opt.iters = double(params.iters); opt.decay_interval = 25; opt.hybrid_strategy = 0; opt.t0 = 0;
    switch class(G)
        case 'ConstantStepSize',  opt.step_kind = 0; opt.lr = G.learning_rate;
        case 'PiecewiseStepSize', opt.step_kind = 1; opt.lr = G.learning_rate; opt.decay_interval = G.decay_interval; opt.t0 = G.t;
        case 'HybridGradient',    opt.step_kind = 2; opt.lr = G.lr; opt.decay_interval = G.decay_interval; opt.hybrid_strategy = G.strategy; opt.t0 = G.t;
        otherwise, error('desc_amd:Gradient', 'params.Gradient must be a ConstantStepSize, PiecewiseStepSize or HybridGradient object');
    end
    opt.seed = 0;   if isfield(params, 'seed'),   opt.seed = params.seed;     end
    opt.device = 0; if isfield(params, 'device'), opt.device = params.device; end
    opt.verbose = 1;
    disp('compute R cycle'); disp('S0Mat'); disp('Initialization completed!'); disp('Reweighting Procedure Started ...')   % DESC.m:132,145,160,162
    [R_est, R_init, S_sorted, info] = desc_amd_mex('desc', int32(IndS - 1), double(RijMat(:,:,perm)), opt);
    disp('DONE!')                                     % DESC.m:313
    if isa(G, 'PiecewiseStepSize') || isa(G, 'HybridGradient'), G.t = info.t_end; end
    S_vec = zeros(1, size(Ind,1));
    S_vec(perm) = S_sorted;
end
