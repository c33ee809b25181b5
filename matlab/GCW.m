%% GCW -- drop-in replacement of the reference's Utils/GCW.m:1 (AdjMat is implied by Ind)
function R_est = GCW(Ind, AdjMat, RijMat, S_vec) %#ok<INUSL>
    [IndS, perm] = sortrows(double(Ind), [1 2]);
    R_est = desc_amd_mex('gcw', int32(IndS - 1), double(RijMat(:,:,perm)), S_vec(perm));
end
