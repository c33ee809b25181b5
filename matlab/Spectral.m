%% Spectral -- drop-in replacement of the reference's Algorithms/Spectral.m:15
function R_est = Spectral(Ind, RijMat)
    [IndS, perm] = sortrows(double(Ind), [1 2]);
    R_est = desc_amd_mex('spectral', int32(IndS - 1), double(RijMat(:,:,perm)));
end
