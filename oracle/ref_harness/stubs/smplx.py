class SMPL:  # absent upstream dependency; fit_smpl itself cannot run here
    def __init__(self, *a, **k):
        raise NotImplementedError("smplx is not available in this image")
