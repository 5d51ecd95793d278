import os, sys
os.environ["CODA_SA_DEBUG"] = "1"
sys.argv = ["x"]
exec(open("tools/diag_model.py").read().replace('os.environ["CODA_SA_MLP"] = "layers"\nrun("layers-SA fused-attn")\nattention_core.attention = attention_ref\nrun("layers-SA torch-attn")', ''))
