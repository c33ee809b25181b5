%% DESC_PGD -- drop-in replacement of the reference's Algorithms/DESC_PGD.m:14
%%   [S_vec] = DESC_PGD(Ind, RijMat, params)
%% Same inputs, same output, same progress lines; the work is done by the MI355X HIP library
%% libdesc_amd.so through the MEX shim desc_pgd_mex (matlab/desc_pgd_mex.c).
%%
%% Ind:    edge_num x 2, rows (i,j), i<j, 1-based            (DESC_PGD.m:5)
%% RijMat: 3 x 3 x edge_num relative rotations               (DESC_PGD.m:7)
%% params: struct with .iters (DESC_PGD.m:170), .Gradient (ConstantStepSize /
%%         PiecewiseStepSize / HybridGradient object, DESC_PGD.m:207), .make_plots (:235).
%%         Optional, not in the reference: .seed (cycle-sampling seed, default 0), .device.
%% Put this directory BEFORE the reference's Algorithms/ on the MATLAB path.
%% params.make_plots = true (:235-239): the library returns mean|ErrVec - S_vec| and GCW(S_vec) of every iteration
%% (desc_pgd_run_traced); the alignment of :238 is the reference's own Utils/GlobalSOdCorrectRight, which must be on the path.
%% The traces the reference keeps in local variables are available as an optional second output:
%%   [S_vec, traces] = DESC_PGD(...)   traces.svec_errors / .MSE_means / .MSE_medians / .obj_vals
function [S_vec, traces] = DESC_PGD(Ind, RijMat, params)
    make_plots = isfield(params, 'make_plots') && params.make_plots;
    % the reference silently requires Ind sorted as (1,2),(1,3),...,(2,3),... (DESC_PGD.m:5,31-34)
    [IndS, perm] = sortrows(double(Ind), [1 2]);
    if ~isequal(perm(:)', 1:size(Ind,1)), RijMat = RijMat(:,:,perm); end

    G = params.Gradient;
    opt.iters = double(params.iters);
    opt.beta1 = 0.9; opt.beta2 = 0.999; opt.decay_interval = 25; opt.hybrid_strategy = 0; opt.t0 = 0;
    switch class(G)
        case 'ConstantStepSize'   % Utils/ConstantStepSize.m
            opt.step_kind = 0; opt.lr = G.learning_rate;
        case 'PiecewiseStepSize'  % Utils/PiecewiseStepSize.m
            opt.step_kind = 1; opt.lr = G.learning_rate; opt.decay_interval = G.decay_interval; opt.t0 = G.t;
        case 'HybridGradient'     % Utils/HybridGradient.m
            opt.step_kind = 2; opt.lr = G.lr; opt.beta1 = G.beta_1; opt.beta2 = G.beta_2;
            opt.decay_interval = G.decay_interval; opt.hybrid_strategy = G.strategy; opt.t0 = G.t;
        otherwise
            error('desc_amd:Gradient', 'params.Gradient must be a ConstantStepSize, PiecewiseStepSize or HybridGradient object');
    end
    opt.seed = 0;   if isfield(params, 'seed'),   opt.seed = params.seed;     end
    opt.device = 0; if isfield(params, 'device'), opt.device = params.device; end
    opt.verbose = 1;   % the per-iteration line of DESC_PGD.m:241 is printed by the MEX shim while the loop runs
    if make_plots
        opt.make_plots = 1;
        ErrVec = double(params.ErrVec(:)');               % DESC_PGD.m:236
        opt.ErrVec = ErrVec(perm);
    end

    disp('compute R cycle')                  % DESC_PGD.m:132
    disp('S0Mat')                            % :145
    disp('Initialization completed!')        % :160
    disp('Reweighting Procedure Started ...')% :162
    adam_m = []; adam_v = [];
    if isa(G, 'HybridGradient') && G.t > 0, adam_m = G.m_t; adam_v = G.v_t; end
    [S_sorted, info] = desc_pgd_mex(int32(IndS - 1), double(RijMat), opt, adam_m, adam_v);
    % handle-object state after the run
    if isa(G, 'PiecewiseStepSize') || isa(G, 'HybridGradient'), G.t = info.t_end; end
    if isa(G, 'HybridGradient') && G.strategy == 0, G.m_t = info.adam_m; G.v_t = info.adam_v; end
    S_vec = zeros(1, size(Ind,1));
    S_vec(perm) = S_sorted;
    traces = struct('obj_vals', info.obj_vals(1:info.iters_run), 'svec_errors', [], 'MSE_means', [], 'MSE_medians', []);
    if make_plots
        traces.svec_errors = info.svec_errors;
        for t = 1:size(info.R_est_all, 4)                 % DESC_PGD.m:238
            [~, traces.MSE_means(end+1), traces.MSE_medians(end+1), ~] = GlobalSOdCorrectRight(info.R_est_all(:,:,:,t), params.R_orig);
        end
    end
end
