"""Head configurations of the heads goldens (shared by make_golden_heads.py and the tests). MLP_PE is absent on purpose: the
reference's MLPRender_PE sizes its first layer for 3 inputs it never concatenates (models/tensorBase.py:115 vs :126-131), so
it raises a shape error for every pos_pe / view_pe."""
HEADS = {"fea": dict(shadingMode="MLP_Fea", view_pe=2, fea_pe=2, pos_pe=0),
         "fea0": dict(shadingMode="MLP_Fea", view_pe=0, fea_pe=3, pos_pe=0),
         "mlp": dict(shadingMode="MLP", view_pe=4, fea_pe=0, pos_pe=0)}
