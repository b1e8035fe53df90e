"""Field / head shapes other than the driver's 16 / 48 / 27 / 6 / 128 (shared by the golden generator and the tests)."""
SHAPES = {
    # upstream TensoRF's default component counts (different per plane), the driver's head
    "upstream": dict(density_n_comp=[16, 4, 4], appearance_n_comp=[48, 12, 12], app_dim=27, shadingMode="MLP_Fea_noview", fea_pe=6,
                     featureC=128, view_pe=6, pos_pe=6),
    # everything smaller: 8 / 24 components, 12 features, two octaves, 64 hidden units
    "small": dict(density_n_comp=[8, 8, 8], appearance_n_comp=[24, 24, 24], app_dim=12, shadingMode="MLP_Fea_noview", fea_pe=2,
                  featureC=64, view_pe=6, pos_pe=6),
    # no feature encoding at all (fea_pe = 0: models/tensorBase.py:104 skips the PE), odd counts
    "nope": dict(density_n_comp=[5, 16, 3], appearance_n_comp=[7, 48, 20], app_dim=9, shadingMode="MLP_Fea_noview", fea_pe=0,
                 featureC=32, view_pe=6, pos_pe=6),
    # a view-dependent head with a narrow hidden layer and the SH head on few components
    "fea64": dict(density_n_comp=[8, 8, 8], appearance_n_comp=[24, 24, 24], app_dim=27, shadingMode="MLP_Fea", fea_pe=2,
                  featureC=64, view_pe=2, pos_pe=6),
    # the parameter-free heads at the kernels' own component counts (their backward: dL/dfeatures from the colour gradients)
    "rgb": dict(density_n_comp=[16, 16, 16], appearance_n_comp=[48, 48, 48], app_dim=3, shadingMode="RGB", fea_pe=6, featureC=128,
                view_pe=6, pos_pe=6),
    "sh16": dict(density_n_comp=[16, 16, 16], appearance_n_comp=[48, 48, 48], app_dim=27, shadingMode="SH", fea_pe=6, featureC=128,
                 view_pe=6, pos_pe=6),
    "sh": dict(density_n_comp=[4, 4, 4], appearance_n_comp=[12, 12, 12], app_dim=27, shadingMode="SH", fea_pe=6, featureC=128,
               view_pe=6, pos_pe=6),
    # LARGER than the tuned kernels hold (round 5: the general-shape path, csrc/t2n_generic.hip): twice the components (different per
    # plane), 256 hidden units; a view-dependent head with 40 features and 192 units; the SH head on 72 components
    "wide": dict(density_n_comp=[32, 20, 24], appearance_n_comp=[96, 64, 72], app_dim=27, shadingMode="MLP_Fea_noview", fea_pe=6,
                 featureC=256, view_pe=6, pos_pe=6),
    "wide_fea": dict(density_n_comp=[24, 24, 24], appearance_n_comp=[64, 64, 64], app_dim=40, shadingMode="MLP_Fea", fea_pe=3,
                     featureC=192, view_pe=2, pos_pe=6),
    "wide_sh": dict(density_n_comp=[40, 8, 8], appearance_n_comp=[72, 72, 72], app_dim=27, shadingMode="SH", fea_pe=6, featureC=128,
                    view_pe=6, pos_pe=6),
}
WIDE = ("wide", "wide_fea", "wide_sh")    # shapes of the general-shape path (no embedding onto the tuned kernels)
