NAME          25FV47
ROWS
 N  R0000
 E  F1X.0
 E  2SF145
 E  2SF089
 E  2SF129
 E  30M00
 E  30M94
 E  30M91
 E  30ATK
 E  30PGK
 E  30G30
 E  30G17
 E  30G44
 E  30G22
 E  30D22
 E  30D30
 E  30121
 E  30F52
 E  30F48
 E  30128
 E  RB017
 E  2RB018
 E  1RB019
 E  2RB020
 E  RC0ST
 L  RCRFT
 E  RMSEX
 E  RLDEX
 E  RPKEX
 E  RRKEX
 E  RG0EX
 E  RF0EX
 E  KW1RC
 E  RA001
 E  RA002
 E  1RA002
 E  RA004
 E  RA005
 E  2RA005
 E  RA006
 E  RA007
 E  RA008
 E  RA009
 E  RA010
 E  RA011
 E  RA012
 E  RA013
 E  RA014
 E  RA015
 E  RA016
 E  RA017
 E  RA018
 E  RA019
 E  RA020
 E  1RA020
 E  RA022
 E  RA023
 E  RA024
 E  RA026
 E  RA027
 E  RA028
 E  RA029
 E  1RA029
 E  RA031
 E  RA032
 E  RA033
 E  RA034
 E  1RA034
 E  2RA034
 E  RA035
 E  RA036
 L  RA037
 E  RA038
 E  RA039
 E  RA040
 E  RA041
 E  RA043
 E  1RA043
 E  RA044
 E  RA045
 E  RA046
 E  RA047
 E  1RA047
 E  RA048
 E  RA049
 E  RA051
 E  RA052
 E  1RA052
 E  1RA053
 E  RA054
 L  RA055
 L  1RA055
 L  RA056
 L  RA057
 L  RA058
 E  1RA058
 L  RA059
 L  RA060
 L  2RA060
 L  3RA060
 L  RA061
 L  RA062
 L  RA063
 L  RA064
 L  RA065
 L  RA066
 L  RA067
 L  RA068
 L  RA069
 L  RA070
 L  RA072
 L  RA075
 L  1RA075
 L  RA076
 L  1RA076
 L  RA077
 L  RA078
 L  1RA078
 L  RA079
 L  1RA079
 L  RA081
 L  1RA081
 L  RA082
 L  RA083
 L  RA084
 L  RA085
 L  RA086
 L  RA087
 L  RA088
 L  RA089
 L  RA090
 L  RA091
 L  RA092
 L  RA093
 L  RA094
 L  RA095
 L  RA096
 E  RA097
 E  1RA099
 L  RA100
 E  RB001
 E  1RB001
 E  RB002
 E  RB003
 E  1RB003
 E  RB004
 E  1RB004
 E  RB005
 E  1RB005
 E  2RB005
 E  RB006
 E  1RB006
 E  RB007
 E  1RB007
 E  2RB007
 E  RB008
 E  1RB008
 E  2RB008
 E  3RB008
 E  RB009
 E  RB010
 E  RB011
 E  RB012
 E  RB013
 E  1RB013
 E  RB014
 E  RB015
 E  1RB015
 E  2RB015
 E  RB016
 E  1RB016
 E  RB018
 E  1RB018
 E  RB019
 E  RB020
 E  1RB020
 E  1RB021
 E  RB022
 E  1RB022
 E  1RB023
 E  RB024
 E  1RB024
 E  RB025
 E  1RB025
 E  2RB025
 E  RB026
 E  RB027
 E  RB028
 E  RB029
 E  RB030
 E  1RB030
 E  2RB030
 E  RB031
 E  1RB031
 E  RB033
 E  RB034
 E  RB035
 E  RB036
 E  RB037
 E  RB038
 E  1RB038
 E  RB039
 E  RB040
 E  1RB040
 E  2RB040
 E  RB041
 E  RB042
 E  1RB043
 E  1RB044
 E  RB045
 E  1RB045
 E  4RB045
 L  RB046
 E  RB047
 E  2RB047
 E  3RB047
 E  RB048
 E  2RB049
 E  RB050
 E  1RB050
 E  RB051
 E  1RB051
 E  2RB051
 E  RB052
 E  RB053
 E  1RB053
 E  RB054
 E  1RB054
 E  RB055
 L  RB057
 L  RB058
 L  1RB058
 L  RB059
 L  RB060
 L  RB061
 L  RB062
 L  RB063
 E  RB064
 L  RB065
 L  RB066
 L  RB067
 E  RB068
 L  RB069
 L  RB070
 L  RB071
 L  RB072
 E  RB073
 L  RB074
 L  RB075
 L  RB076
 L  RB077
 L  RB078
 L  RB079
 L  RB080
 L  RB081
 L  RB082
 L  RB083
 L  RB084
 L  RB085
 L  RB086
 L  RB087
 L  1RB087
 L  RB088
 L  RB089
 L  RB090
 L  1RB090
 L  RB091
 L  RB092
 L  1RB092
 L  RB093
 L  1RB093
 L  RB094
 L  RB095
 L  RB096
 L  RB097
 L  RB098
 L  RB099
 L  RB100
 L  RB101
 L  RB102
 L  RB103
 L  RB104
 L  RB105
 L  RB106
 E  RCLD4
 E  RCF35
 E  RCF80
 E  RDBMS
 E  RDLD4
 E  RDK19
 E  RDG55
 E  RDF35
 E  RE001
 E  RE002
 E  RE004
 L  RE005
 E  RE008
 E  RE009
 E  RE010
 E  RE011
 E  RE012
 E  RE013
 E  RE014
 E  RE015
 E  RE016
 E  RE017
 E  RE018
 E  RE019
 E  RE021
 E  RE022
 E  RE024
 E  RE025
 E  RE026
 E  RE028
 E  RE030
 E  RE035
 E  RE036
 E  RE037
 E  RE039
 E  RE040
 E  RE044
 E  RE045
 E  RE046
 E  RE047
 E  RE048
 E  RE049
 E  RE050
 E  RE051
 E  RE055
 E  1RE056
 E  RE057
 E  RE058
 E  RE060
 L  RE061
 L  1RE061
 L  2RE061
 L  RE062
 L  RE063
 L  RE064
 L  RE066
 L  RE068
 L  RE069
 L  RE070
 L  RE071
 L  RE072
 L  RE073
 L  RE074
 L  RE075
 L  RE084
 L  RE085
 L  RE086
 L  RE087
 L  RE088
 L  RE089
 L  RE090
 L  RE091
 L  RE092
 L  RE098
 L  RE099
 L  RE100
 L  RE101
 L  RE102
 E  RE103
 L  RE104
 L  RH001
 E  1RH001
 E  2RH001
 E  RH002
 E  RH003
 E  RH004
 E  RH005
 E  1RH005
 E  RH006
 E  1RH006
 E  RH007
 E  RH008
 E  1RH008
 E  RH009
 E  RH010
 E  RH011
 E  RH012
 E  RH013
 E  RH014
 E  RH015
 E  RH016
 E  RH017
 E  1RH017
 E  2RH017
 E  RH018
 E  1RH018
 E  2RH018
 E  3RH018
 E  RH019
 E  1RH019
 E  RH020
 E  RH021
 E  1RH021
 E  RH022
 E  1RH022
 E  2RH022
 E  3RH022
 E  4RH022
 E  RH023
 E  2RH023
 E  RH024
 E  1RH024
 E  RH025
 E  1RH025
 E  2RH025
 E  RH026
 E  1RH026
 E  2RH026
 E  RH027
 E  1RH027
 E  RH028
 E  1RH028
 E  2RH028
 E  RH029
 E  1RH029
 E  RH030
 E  1RH030
 E  2RH030
 E  3RH030
 E  4RH030
 E  RH031
 E  RH032
 E  1RH033
 E  2RH033
 E  RH034
 E  1RH034
 E  RH035
 E  1RH035
 E  RH036
 E  1RH036
 E  RH038
 E  1RH038
 E  2RH038
 E  RH039
 E  2RH039
 E  RH041
 E  RH042
 E  2RH042
 E  RH043
 E  2RH043
 E  RH046
 E  RH048
 E  2RH048
 E  3RH048
 E  RH050
 E  RH051
 E  1RH051
 E  2RH051
 E  RH052
 E  2RH052
 E  3RH052
 E  4RH052
 E  RH053
 E  2RH053
 E  3RH053
 L  RH054
 L  1RH054
 L  RH055
 L  RH056
 L  RH057
 L  RH058
 L  RH059
 L  RH060
 L  1RH060
 L  RH061
 L  RH062
 E  RH063
 L  RH064
 L  RH065
 L  RH066
 L  1RH066
 E  RH067
 L  RH068
 L  RH069
 L  RH070
 L  RH071
 E  RH072
 L  RH073
 L  RH074
 L  RH075
 L  RH080
 L  RH081
 L  RH082
 L  RH083
 L  RH084
 L  RH085
 L  RH086
 L  RH087
 L  RH088
 L  RH089
 L  1RH089
 L  2RH089
 L  3RH089
 L  RH090
 L  1RH090
 L  2RH090
 L  RH091
 L  RH093
 L  1RH093
 L  RH094
 E  RH095
 E  RH096
 L  RH099
 L  RH103
 L  RH105
 L  RH106
 L  1RH106
 L  2RH106
 L  RH107
 L  1RH107
 L  2RH107
 L  RH108
 E  RH109
 L  1RH109
 L  2RH109
 L  RH110
 E  RH113
 E  RH114
 L  RH115
 L  RH116
 L  RH117
 L  RH118
 L  1RH118
 E  R1002
 E  R1003
 E  R1004
 E  1R1004
 E  R1005
 E  1R1006
 L  2R1006
 E  R1007
 E  1R1007
 E  R1008
 E  R1009
 L  1R1009
 E  R1010
 E  R1012
 E  R1013
 E  R1014
 E  R1015
 E  R1016
 E  R1017
 E  R1018
 E  R1019
 E  1R1019
 E  R1020
 E  1R1020
 E  R1021
 E  1R1021
 E  2R1021
 E  3R1021
 E  R1022
 E  R1023
 E  R1024
 E  R1025
 E  R1026
 E  R1027
 E  1R1027
 E  R1028
 E  1R1028
 E  2R1028
 E  3R1028
 E  R1029
 E  R1030
 E  R1031
 E  R1032
 E  R1033
 E  1R1033
 E  2R1033
 E  3R1033
 E  R1034
 E  R1035
 E  R1036
 E  R1037
 E  R1038
 E  1R1038
 E  2R1038
 E  R1039
 L  1R1039
 E  R1040
 E  R1041
 L  R1042
 E  R1043
 E  R1045
 E  R1046
 E  R1047
 E  R1050
 E  R1051
 E  R1054
 E  R1055
 E  R1057
 L  1R1057
 E  R1058
 E  R1060
 E  R1061
 E  R1062
 E  2R1062
 E  R1063
 E  1R1063
 E  2R1063
 E  R1064
 E  1R1064
 E  R1065
 E  R1066
 E  R1068
 E  1R1068
 E  R1069
 L  R1070
 L  R1071
 L  1R1071
 L  2R1071
 L  R1072
 L  R1073
 L  R1074
 E  R1075
 L  R1076
 L  R1077
 L  R1078
 E  R1079
 L  R1080
 L  R1081
 L  R1082
 L  R1086
 L  R1087
 L  R1088
 L  R1089
 L  R1090
 L  R1091
 L  R1092
 L  R1093
 L  R1094
 L  R1095
 L  R1096
 L  R1097
 L  R1098
 L  R1099
 L  R1100
 L  R1101
 L  R1102
 L  1R1102
 L  R1103
 L  R1104
 L  R1105
 L  R1106
 L  R1107
 L  R1108
 L  R1109
 L  R1110
 L  R1111
 L  R1112
 L  R1113
 L  R1114
 L  R1115
 E  RJ001
 E  1RJ001
 E  RJ002
 E  RJ003
 E  RJ004
 E  RJ005
 E  RJ006
 E  1RJ006
 E  RJ007
 E  1RJ007
 E  RJ008
 E  1RJ008
 E  RJ009
 E  RJ010
 E  RJ011
 E  1RJ011
 E  RJ012
 E  RJ013
 E  RJ014
 E  RJ015
 E  RJ016
 E  RJ018
 E  RJ019
 E  RJ020
 E  RJ021
 E  RJ024
 E  RJ025
 E  RJ026
 E  RJ027
 E  RJ028
 E  RJ029
 E  RJ030
 E  RJ031
 E  RJ032
 E  RJ033
 L  RJ034
 E  RJ035
 E  RJ036
 E  1RJ036
 E  1RJ037
 L  2RJ037
 E  RJ038
 E  RJ039
 E  RJ040
 E  1RJ041
 E  RJ042
 E  RJ043
 E  RJ045
 E  2RJ045
 E  RJ046
 E  2RJ046
 E  RJ047
 L  RJ048
 L  RJ049
 L  RJ050
 L  1RJ050
 L  RJ051
 E  RJ052
 L  RJ053
 L  RJ054
 E  RJ055
 L  RJ056
 L  RJ057
 L  RJ061
 L  RJ062
 L  RJ063
 L  RJ064
 L  RJ065
 L  RJ066
 L  RJ067
 L  RJ068
 L  RJ069
 L  RJ070
 L  RJ071
 L  RJ072
 L  RJ073
 L  1RJ073
 L  RJ074
 L  1RJ074
 L  RJ075
 L  RJ076
 L  1RJ076
 L  RJ077
 L  RJ078
 L  RJ079
 L  RJ080
 E  RJ081
 E  RJ082
 E  RJ083
 E  RJ084
 E  1RJ084
 E  RJ089
 E  RJ091
 E  RJ092
 L  RJ095
 L  RJ096
 E  RJ097
 E  1RJ097
 E  RJ098
 E  1RJ098
 L  RJ099
 L  RJ100
 L  RJ102
 L  RJ103
 L  RJ104
 L  RJ105
 L  RJ106
 L  1RM001
 E  RM002
 E  1RM003
 E  2RM003
 E  RM004
 E  RM005
 E  RM006
 E  1RM006
 E  2RM006
 E  RM011
 E  RM012
 E  RN009
 E  1RN014
 E  RR010
 E  RR021
 E  RRG29
 L  UKLDF
 L  HSPMS
 L  HSRMS
 L  HSLDF
 L  HSATK
 L  NSG17
 L  HSG58
 L  HSF80
 E  RT005
 E  RT006
 E  3RT006
 E  4RT006
 E  RV001
 E  RV002
 E  RV003
 E  RXATG
 E  RXF72
 E  RY001
 E  1RY001
 E  RY002
 E  RY004
 L  RY005
 E  RZ001
 E  RZ002
 E  RZ004
 E  RZ005
 E  RZ006
 E  RZ012
 E  RZ016
 E  RZ019
 E  RZ020
 L  RABDH
 E  RAGJA
 L  RGACH
 L  RHN1G
 L  RK1RK
 E  RKUWT
 L  RL1BY
 L  RMN1G
 L  RMURB
 L  RQATR
 L  RT1JA
COLUMNS
    MSEXP     R0000              -.1   RMSEX              -1.
    MS1MP     R0000             .101   RMSEX               1.
    LDEXP     R0000            -.075   RLDEX              -1.
    LD1MP     R0000             .076   RLDEX               1.
    PKEXP     R0000              -.1   RPKEX              -1.
    PK1MP     R0000             .101   RPKEX               1.
    RKEXP     R0000           -.0875   RRKEX              -1.
    RK1MP     R0000            .0885   RRKEX               1.
    1G0EXP    RG0EX              -1.
    1G01MP    RG0EX               1.
    1F0EXP    RF0EX              -1.
    1F01MP    RF0EX               1.
    5C0ST     R0000               .2   RC0ST              -1.
    6C0ST     R0000              .25   RC0ST              -1.
    CCRFT     R0000             .001   RCRFT              -1.
    CRUDE     R0000            2.275   RKUWT              -1.
    AABDH     R0000          -1.5303   1RA002             -1.
    AABDH     RABDH               1.
    HABDH     R0000            .9632   RH006              -1.
    HABDH     RABDH               1.
    JABDH     R0000            .3687   RJ006              -1.
    JABDH     RABDH               1.
    1AAGJA    R0000          -1.4254   RA001              -1.
    1AAGJA    RAGJA               1.
    1BAGJA    R0000            1.007   RB001              -1.
    1BAGJA    RAGJA               1.
    EAGJA     R0000          -1.1387   RE002              -1.
    EAGJA     RAGJA               1.
    1HAGJA    R0000            .9839   RH002              -1.
    1HAGJA    RAGJA               1.
    11AGJA    R0000           1.0109   R1004              -1.
    11AGJA    RAGJA               1.
    1JAGJA    R0000            .3969   RJ002              -1.
    1JAGJA    RAGJA               1.
    AGACH     R0000          -1.2464   RA004              -1.
    AGACH     RGACH               1.
    BGACH     R0000            1.007   RB003              -1.
    BGACH     RGACH               1.
    EGACH     R0000          -1.0122   RE004              -1.
    EGACH     RGACH               1.
    1GACH     R0000           1.0109   R1005              -1.
    1GACH     RGACH               1.
    JGACH     R0000            .3969   RJ005              -1.
    JGACH     RGACH               1.
    BHN1G     R0000            .8625   RB004              -1.
    BHN1G     RHN1G               1.
    HHN1G     R0000            .6674   RH005              -1.
    HHN1G     RHN1G               1.
    1HN1G     R0000            .6943   R1007              -1.
    1HN1G     RHN1G               1.
    BK1RK     R0000            .3075   1RB001             -1.
    BK1RK     1RB038            -.02   RB042             -.02
    BK1RK     RB047             -.02   RB052              .06
    BK1RK     RK1RK               1.
    HK1RK     R0000            .4249   RH003              -1.
    HK1RK     RH039             -.02   RH042             -.02
    HK1RK     RH048             -.02   3RH052             .06
    HK1RK     RK1RK               1.
    1K1RK     R0000            .3059   R1002              -1.
    1K1RK     R1055             -.02   R1057             -.02
    1K1RK     1R1057            -.02   R1062             -.02
    1K1RK     R1068              .06   RK1RK               1.
    AKUWT     R0000          -1.2063   RA005              -1.
    AKUWT     RKUWT               1.
    BKUWT     R0000           1.0114   RB002              -1.
    BKUWT     RKUWT               1.
    EKUWT     R0000           -.9791   RE001              -1.
    EKUWT     RKUWT               1.
    HKUWT     R0000            .9886   1RH001             -1.
    HKUWT     RKUWT               1.
    1KUWT     R0000           1.0154   R1003              -1.
    1KUWT     RKUWT               1.
    JKUWT     R0000            .4033   RJ001              -1.
    JKUWT     RKUWT               1.
    BL1BY     R0000            .3657   1RB003             -1.
    BL1BY     RL1BY               1.
    HL1BY     R0000             .345   1RH006             -1.
    HL1BY     RL1BY               1.
    1L1BY     R0000            .6832   1R1006             -1.
    1L1BY     RL1BY               1.
    BMN1G     R0000            .8625   1RB004             -1.
    BMN1G     RMN1G               1.
    HMN1G     R0000            .6674   1RH005             -1.
    HMN1G     RMN1G               1.
    1MN1G     R0000            .6943   1R1004             -1.
    1MN1G     RMN1G               1.
    1MURB     R0000           1.0199   1R1007             -1.
    1MURB     RMURB               1.
    JMURB     R0000            .4094   1RJ006             -1.
    JMURB     RMURB               1.
    AQATR     R0000          -1.6519   RA002              -1.
    AQATR     RQATR               1.
    JQATR     R0000            .4001   RJ004              -1.
    JQATR     RQATR               1.
    ASPKU     R0000          -1.3823   2RA005             -1.
    ASPKU     RY001              .15   RKUWT              .85
    HSPKU     R0000            .9886   2RH001             -1.
    HSPKU     RY001               .1   RKUWT               .9
    JSPKU     R0000            .4033   1RJ001             -1.
    JSPKU     RY001               .1   RKUWT               .9
    BT1JA     R0000            .4183   RB005              -1.
    BT1JA     RT1JA               1.
    HT1JA     R0000            .4022   RH004              -1.
    HT1JA     RT1JA               1.
    HC3       R0000             100.   1RH008              1.
    AVGAS     R0000          20.8248   RC0ST            1.658
    AVGAS     RMSEX              -1.   RH031               1.
    KWPMS     R0000              20.   RA039               1.
    PMS       R0000          13.9748   RC0ST            1.658
    PMS       RMSEX              -1.   1RH035              1.
    RMS       R0000          11.5968   RC0ST            1.658
    RMS       RMSEX              -1.   1RH036              1.
    PGB0      R0000          12.0848   RC0ST            1.658
    PGB0      RPKEX              -1.   RH041               1.
    ATK       R0000          11.1467   RC0ST            1.658
    ATK       RRKEX              -1.   RH042               1.
    KWG20     R0000              15.   RA044               1.
    G29       R0000           7.9067   RC0ST            1.646
    G29       RRG29               1.
    G44       R0000           9.5828   RC0ST            1.658
    G44       RG0EX              -1.   RH043               1.
    G44       RH082              -5.
    G58       R0000           8.0198   RC0ST            1.658
    G58       RG0EX              -1.   2RH048              1.
    F28       R0000           5.9417   RC0ST            1.658
    F28       RF0EX              -1.   1RH051              1.
    F80       R0000           4.6059   RC0ST            1.658
    F80       RF0EX              -1.   2RH052              1.
    F81       R0000            5.535   RC0ST            1.562
    F81       RF0EX              -1.   RB054               1.
    ADDAE     R0000            .6374   KW1RC               1.
    CA001     R0000            .0174   1RA002          1.0125
    CA001     RA009             .167   RA011              .06
    CA001     RA012             .045   RA013               .1
    CA001     RA014              .09   RA016             .116
    CA001     1RA020            .404   RA035             .018
    CA001     RA036           -.0223   RA055             1.13
    CA001     1RA055            1.13
    CA002     R0000            .0111   RA001           1.0115
    CA002     RA008             .037   RA009             .151
    CA002     RA010             .053   RA013             .088
    CA002     RA014              .08   RA016             .104
    CA002     RA018             .467   RA035              .02
    CA002     RA036           -.0217   RA055             1.02
    CA002     1RA055            1.02
    CA003     R0000            .0088   RA005           1.0085
    CA003     RA009             .131   RA011             .042
    CA003     RA012             .033   RA013             .076
    CA003     RA015             .091   RA017              .06
    CA003     RA019             .546   RA035             .021
    CA003     RA036           -.0223   RA055               1.
    CA003     RA072            -.101
    CA004     R0000            .0174   RA002           1.0125
    CA004     RA009             .192   RA011              .07
    CA004     RA012             .048   RA013             .115
    CA004     RA014             .113   RA016             .074
    CA004     RA020             .355   RA035             .033
    CA004     RA036           -.0223   RA055             1.25
    CA006     R0000            .0114   RA004           1.0115
    CA006     RA006             .533   RA008             .037
    CA006     RA009             .147   RA010             .037
    CA006     RA013              .08   RA014             .074
    CA006     RA016             .073   RA035             .019
    CA006     RA036           -.0223   RA055               1.
    CA006     1RA055              1.
    CA008     R0000            .2564   RA061           -10.52
    CA008     RA062               1.
    CA009     R0000            .2564   RA066           -15.27
    CA009     RA067               1.
    CA010     R0000            .0819   RA035              -1.
    CA010     RA039               1.   RA061             5.17
    CA010     RA062            -.541   RA063      -155.159897
    CA010     RA064       218.259996   RA065        95.509999
    CA011     R0000            .0495   RA035              -1.
    CA011     RA040               1.   RA066           -18.96
    CA011     RA067            -.734   RA068      -155.159897
    CA011     RA069       218.259996   RA070        95.509999
    CA012     R0000            .0404   RA009              -1.
    CA012     RA040               1.   RA066            33.79
    CA012     RA067            -.599   RA068           -14.08
    CA012     RA069           -14.78   RA070           -6.899
    CA013     R0000            .1593   RA009              -1.
    CA013     RA013              .02   RA022             .149
    CA013     RA035             .241   RA036            -.115
    CA013     RA039              .59   RA056            1.405
    CA013     RA061              .39   RA062            -.242
    CA013     RA063           -3.855   RA064             -4.7
    CA013     RA065            -3.55
    CA014     R0000            .0543   RA009              -1.
    CA014     RA013              .02   RA022             .149
    CA014     RA035             .112   RA036            -.082
    CA014     RA040             .719   RA056            1.117
    CA014     RA066             -.78   RA067            -.414
    CA014     RA068             .972   RA069           -13.22
    CA014     RA070           -4.471
    CA015     R0000            .1001   RA009              -1.
    CA015     RA013              .02   RA023             .446
    CA015     RA035             .153   RA036            -.074
    CA015     RA039             .381   RA056             .801
    CA015     RA061              .24   RA062            -.153
    CA015     RA063             7.32   RA064              -4.
    CA015     RA065            -2.24
    CA016     R0000            .0382   RA009              -1.
    CA016     RA013              .02   RA023             .446
    CA016     RA035             .043   RA036           -.0527
    CA016     RA040             .491   RA056             .879
    CA016     RA066             -.52   RA067            -.277
    CA016     RA068            16.25   RA069            -10.4
    CA016     RA070            -2.99
    CA017     R0000            .1087   RA009              -1.
    CA017     RA013              .02   RA022             .149
    CA017     RA035             .191   RA036           -.1099
    CA017     RA039              .64   RA056            1.245
    CA017     RA061             3.73   RA062            -.266
    CA017     RA063           -1.696   RA064            -8.06
    CA017     RA065             -3.9
    CA018     R0000            .0928   RA009              -1.
    CA018     RA013              .02   RA022             .149
    CA018     RA035             .191   RA036           -.1099
    CA018     RA040              .64   RA056            1.245
    CA018     RA066            -7.97   RA067            -.361
    CA018     RA068           -1.696   RA069            -8.06
    CA018     RA070             -3.9
    CA019     R0000             .071   RA009              -1.
    CA019     RA013              .02   RA023             .446
    CA019     RA035             .083   RA036           -.0706
    CA019     RA039             .451   RA056             .939
    CA019     RA061             2.58   RA062            -.184
    CA019     RA063            11.72   RA064            -6.91
    CA019     RA065           -2.695
    CA020     R0000              .06   RA009              -1.
    CA020     RA013              .02   RA023             .446
    CA020     RA035             .083   RA036           -.0706
    CA020     RA040             .451   RA056             .939
    CA020     RA066            -5.51   RA067            -.249
    CA020     RA068            11.72   RA069            -6.91
    CA020     RA070           -2.695
    CA021     R0000            .1724   RA009            -.745
    CA021     RA010            -.255   RA013              .02
    CA021     RA022             .083   RA035             .207
    CA021     RA036            -.124   RA039              .69
    CA021     RA056            1.447   RA061              .45
    CA021     RA062            -.279   RA063             3.56
    CA021     RA064            -8.72   RA065            -4.09
    CA022     R0000             .065   RA009            -.745
    CA022     RA010            -.255   RA013              .02
    CA022     RA022             .083   RA035             .069
    CA022     RA036            -.088   RA040             .828
    CA022     RA056            1.068   RA066             -.88
    CA022     RA067             -.47   RA068            8.832
    CA022     RA069            -20.2   RA070           -5.078
    CA024     R0000            .1417   RA009            -.745
    CA024     RA010            -.255   RA013              .02
    CA024     RA023              .25   RA035             .125
    CA024     RA036            -.101   RA039             .605
    CA024     RA056            1.162   RA061              .38
    CA024     RA062             -.24   RA063           20.682
    CA024     RA064            -9.12   RA065            -3.52
    CA025     R0000             .053   RA009            -.745
    CA025     RA010            -.255   RA013              .02
    CA025     RA023              .25   RA035             .039
    CA025     RA036            -.072   RA040             .691
    CA025     RA056             .858   RA066             -.72
    CA025     RA067            -.384   RA068             29.8
    CA025     RA069           -18.69   RA070           -4.153
    CA027     R0000            .1138   RA009            -.747
    CA027     RA011            -.253   RA013              .02
    CA027     RA022               .1   RA035             .185
    CA027     RA036            -.116   RA039             .695
    CA027     RA056            1.239   RA061               4.
    CA027     RA062            -.285   RA063            3.632
    CA027     RA064           -11.26   RA065           -4.177
    CA028     R0000            .0968   RA009            -.747
    CA028     RA011            -.253   RA013              .02
    CA028     RA022               .1   RA035             .185
    CA028     RA036            -.116   RA040             .695
    CA028     RA056            1.239   RA066            -8.54
    CA028     RA067            -.386   RA068            3.632
    CA028     RA069           -11.26   RA070           -4.177
    CA029     R0000            .1033   RA009            -.747
    CA029     RA011            -.253   RA013              .02
    CA029     RA022               .1   RA035             .185
    CA029     RA036            -.131   RA039             .347
    CA029     RA040             .348   RA056            1.239
    CA029     RA061            -2.07   RA062            -.129
    CA029     RA063           18.585   RA064           -12.06
    CA029     RA065           -3.965   RA066             2.96
    CA029     RA067             -.21   RA068            -7.26
    CA029     RA069            -1.63   RA070             -.79
    CA030     R0000            .0977   RA009            -.745
    CA030     RA010            -.255   RA013              .02
    CA030     RA023              .25   RA035             .088
    CA030     RA036           -.0966   RA039             .642
    CA030     RA056               1.   RA061             3.62
    CA030     RA062            -.258   RA063            23.86
    CA030     RA064           -12.67   RA065           -3.786
    CA031     R0000            .0823   RA009            -.745
    CA031     RA010            -.255   RA013              .02
    CA031     RA023              .25   RA035             .088
    CA031     RA036           -.0966   RA040             .642
    CA031     RA056               1.   RA066            -7.74
    CA031     RA067             -.35   RA068            23.86
    CA031     RA069           -12.67   RA070           -3.786
    CA032     R0000            .0976   RA009            -.745
    CA032     RA010            -.255   RA013              .02
    CA032     RA023              .25   RA035             .088
    CA032     RA036           -.1086   RA039             .321
    CA032     RA040             .321   RA056               1.
    CA032     RA061            -1.88   RA062            -.118
    CA032     RA063           20.625   RA064           -10.95
    CA032     RA065             -3.6   RA066             2.23
    CA032     RA067             -.19   RA068            -2.23
    CA032     RA069            -3.08   RA070            -.714
    CA033     R0000            .1655   RA009            -.747
    CA033     RA011            -.253   RA013              .02
    CA033     RA022               .1   RA035             .249
    CA033     RA036            -.122   RA039             .631
    CA033     RA056            1.438   RA061              .41
    CA033     RA062            -.255   RA063            2.442
    CA033     RA064             -5.7   RA065            -3.74
    CA034     R0000            .0621   RA009            -.747
    CA034     RA011            -.253   RA013              .02
    CA034     RA022               .1   RA035             .099
    CA034     RA036            -.087   RA040             .781
    CA034     RA056            1.062   RA066             -.83
    CA034     RA067            -.443   RA068            7.287
    CA034     RA069           -18.43   RA070           -4.789
    CA035     R0000            .1291   RA009            -.747
    CA035     RA011            -.253   RA013              .02
    CA035     RA023               .3   RA035             .139
    CA035     RA036            -.094   RA039             .541
    CA035     RA056            1.172   RA061              .34
    CA035     RA062            -.215   RA063           17.125
    CA035     RA064            -7.54   RA065            -3.15
    CA036     R0000            .0491   RA009            -.747
    CA036     RA011            -.253   RA013              .02
    CA036     RA023               .3   RA035             .045
    CA036     RA036           -.0671   RA040             .635
    CA036     RA056             .872   RA066             -.66
    CA036     RA067            -.353   RA068            25.73
    CA036     RA069           -16.68   RA070           -3.818
    CA037     R0000            .1181   RA009            -.745
    CA037     RA010            -.255   RA013              .02
    CA037     RA022             .083   RA035             .162
    CA037     RA036            -.119   RA039             .735
    CA037     RA056            1.246   RA061             4.23
    CA037     RA062            -.301   RA063            4.805
    CA037     RA064           -12.49   RA065           -4.421
    CA038     R0000               .1   RA009            -.745
    CA038     RA010            -.255   RA013              .02
    CA038     RA022             .083   RA035             .162
    CA038     RA036            -.119   RA040             .735
    CA038     RA056            1.246   RA066            -9.03
    CA038     RA067            -.409   RA068            4.805
    CA038     RA069           -12.59   RA070           -4.421
    CA039     R0000             .118   RA009            -.745
    CA039     RA010            -.255   RA013              .02
    CA039     RA022             .083   RA035             .162
    CA039     RA036            -.133   RA039             .367
    CA039     RA040             .368   RA056            1.246
    CA039     RA061            -2.18   RA062            -.137
    CA039     RA063            19.62   RA064           -12.73
    CA039     RA065           -4.186   RA066             3.13
    CA039     RA067            -.222   RA068            -7.83
    CA039     RA069            -2.35   RA070            -.835
    CA040     R0000            .0906   RA009            -.747
    CA040     RA011            -.253   RA013              .02
    CA040     RA023               .3   RA035              .09
    CA040     RA036             -.09   RA039              .59
    CA040     RA056               1.   RA061             3.33
    CA040     RA062            -.237   RA063           20.412
    CA040     RA064           -11.19   RA065           -3.478
    CA041     R0000            .0764   RA009            -.747
    CA041     RA011            -.253   RA013              .02
    CA041     RA023               .3   RA035              .09
    CA041     RA036             -.09   RA040              .59
    CA041     RA056               1.   RA066            -7.11
    CA041     RA067            -.322   RA068           20.412
    CA041     RA069           -11.19   RA070           -3.478
    CA042     R0000            .0905   RA009            -.747
    CA042     RA011            -.253   RA013              .02
    CA042     RA023               .3   RA035              .09
    CA042     RA036           -.1011   RA039             .295
    CA042     RA040             .295   RA056               1.
    CA042     RA061            -1.73   RA062            -.108
    CA042     RA063           18.975   RA064           -10.07
    CA042     RA065           -3.312   RA066             2.05
    CA042     RA067            -.174   RA068            -2.05
    CA042     RA069            -2.34   RA070            -.656
    CA043     RA010              -1.   RA043               1.
    CA043     RA075            -9.27   RA076             7.82
    CA043     RA077             -3.5   RA078            33.89
    CA043     RA079             -16.
    1CA043    RA010              -1.   RA044               1.
    1CA043    1RA075           -9.27   1RA076            7.82
    1CA043    1RA078           33.89   1RA079            -16.
    CA044     RA010              -1.   RA046               1.
    CA044     RA082           -13.98   RA083            10.15
    CA044     RA084            33.89   RA085            -13.5
    CA044     RA086             -20.   RA087             -14.
    CA044     RA088            -21.5
    CA045     R0000            .0095   RA009              .27
    CA045     RA011             -.42   RA012             -.33
    CA045     RA013             -.52   RA041               1.
    CA045     RA072               .5
    1CA045    R0000            .0095   RA009             -.07
    1CA045    RA011             -.29   RA012             -.23
    1CA045    RA013             -.41   RA041               1.
    1CA045    RA055               .6   RA072               .5
    CA046     RA011              -1.   RA043               1.
    CA046     RA075            -9.27   RA076             7.82
    CA046     RA077             -3.5   RA078            33.89
    CA046     RA079             -16.
    1CA046    RA011              -1.   RA044               1.
    1CA046    1RA075           -9.27   1RA076            7.82
    1CA046    1RA078           33.89   1RA079            -16.
    CA048     RA011              -1.   RA046               1.
    CA048     RA082           -13.98   RA083            10.15
    CA048     RA084            33.89   RA085            -13.5
    CA048     RA086             -20.   RA087             -14.
    CA048     RA088            -21.5
    CA049     RA012              -1.   RA013               1.
    1CA050    R0000            .0104   RA012             -.44
    1CA050    RA013             -.65   RA015              .09
    1CA050    1RA043              1.
    CA051     R0000            .0218   RA013              -1.
    CA051     RA036           -.0409   RA044             .997
    CA051     RA059             .933   1RA075           -4.76
    CA051     1RA076            3.31   1RA078             2.3
    CA051     1RA079          -11.96
    CA052     RA013              -1.   RA043               1.
    CA052     RA075            -4.46   RA076             3.01
    CA052     RA077              -2.   RA078              2.3
    CA052     RA079             -12.
    CA053     RA013              -1.   RA046               1.
    CA053     RA082            -9.15   RA083             5.32
    CA053     RA084              2.3   RA085             -12.
    CA053     RA086             -16.   RA087             -14.
    CA053     RA088             -19.
    CA054     RA013              -1.   RA048               1.
    CA054     RA089           -26.96   RA090            26.15
    CA054     RA091             -22.   RA092             -69.
    CA055     RA013              -1.   RA051               1.
    CA055     RA093           -26.96   RA094            26.15
    CA055     RA095            -30.5   RA096             -60.
    CA056     R0000            .0218   RA014              -1.
    CA056     RA036           -.0409   RA044             .987
    CA056     RA059               1.   1RA075           -.286
    CA056     1RA076           -1.14   1RA078           -3.13
    CA056     1RA079           -.987
    CA057     RA014              -1.   RA043               1.
    CA057     RA075              .71   RA076            -2.16
    CA057     RA077             2.75   RA078            -3.17
    CA057     RA079              -1.
    CA058     RA014              -1.   RA046               1.
    CA058     RA082            -3.98   RA083              .15
    CA058     RA084            -3.17   RA085            -7.25
    CA058     RA086              -5.   RA087             -14.
    CA058     RA088             -15.
    CA059     RA014              -1.   RA048               1.
    CA059     RA089           -19.06   RA090            18.25
    CA059     RA091           -17.25   RA092             -53.
    CA060     RA014              -1.   RA051               1.
    CA060     RA093           -19.06   RA094            18.25
    CA060     RA095           -25.75   RA096             -44.
    CA061     R0000            .0218   RA015              -1.
    CA061     RA036           -.0409   RA044              .98
    CA061     RA059            1.005   1RA075           -1.57
    CA061     1RA076             .14   1RA078           -3.19
    CA061     1RA079           -1.47
    CA062     RA015              -1.   RA043               1.
    CA062     RA076            -1.45   RA077              6.5
    CA062     RA078            -3.26   RA079             -1.5
    CA063     RA015              -1.   RA046               1.
    CA063     RA082            -4.69   RA083              .86
    CA063     RA084            -3.26   RA085             -3.5
    CA063     RA086             -5.5   RA087             -14.
    CA063     RA088             -16.
    CA064     RA015              -1.   RA048               1.
    CA064     RA089           -18.96   RA090            18.15
    CA064     RA091            -13.5   RA092             -54.
    CA065     RA015              -1.   RA051               1.
    CA065     RA093           -18.96   RA094            18.15
    CA065     RA095             -22.   RA096             -45.
    CA066     R0000           -.0027   RA016              -1.
    CA066     RA026               1.   RA032            -.054
    CA066     RA034             .023   RA035             .031
    CA068     R0000            .0218   RA016              -1.
    CA068     RA036           -.0409   RA044             .976
    CA068     RA059            1.835   1RA075            2.33
    CA068     1RA076           -3.75   1RA078           -3.51
    CA068     1RA079            20.5
    CA069     RA016              -1.   RA043               1.
    CA069     RA075             4.15   RA076             -5.6
    CA069     RA077              8.5   RA078             -3.6
    CA069     RA079              21.
    CA070     RA016              -1.   RA046               1.
    CA070     RA082             -.54   RA083            -3.29
    CA070     RA084             -3.6   RA085             -1.5
    CA070     RA086              17.   RA087             -14.
    CA070     RA088              -9.
    CA071     RA016              -1.   RA048               1.
    CA071     RA089           -12.36   RA090            11.55
    CA071     RA091            -11.5   RA092             -30.
    CA072     RA016              -1.   RA051               1.
    CA072     RA093           -12.36   RA094            11.55
    CA072     RA095             -20.   RA096             -21.
    CA073     R0000           -.0027   RA017              -1.
    CA073     RA026               1.   RA032            -.054
    CA073     RA034             .023   RA035             .031
    CA075     R0000            .0218   RA017              -1.
    CA075     RA036           -.0409   RA044             .964
    CA075     RA059            1.835   1RA075            6.63
    CA075     1RA076           -2.77   1RA078           -3.46
    CA075     1RA079           25.06
    CA076     RA017              -1.   RA043               1.
    CA076     RA075             4.15   RA076             -5.6
    CA076     RA077             14.5   RA078            -3.59
    CA076     RA079              26.
    CA077     RA017              -1.   RA046               1.
    CA077     RA082             -.54   RA083            -3.29
    CA077     RA084            -3.59   RA085              4.5
    CA077     RA086              22.   RA087             -14.
    CA077     RA088              -6.
    CA078     RA017              -1.   RA048               1.
    CA078     RA089           -11.56   RA090            10.75
    CA078     RA091             -5.5   RA092             -25.
    CA079     RA017              -1.   RA051               1.
    CA079     RA093           -11.56   RA094            10.75
    CA079     RA095             -14.   RA096             -16.
    CA080     R0000            .0075   RA018              -1.
    CA080     RA024             .081   RA026             .508
    CA080     RA027             .401   RA034              .01
    CA080     RA036           -.0029   RA057               1.
    CA081     RA018              -1.   RA048               1.
    CA081     RA089             3.44   RA090            -4.25
    CA081     RA091               .5   RA092              15.
    CA082     RA018              -1.   RA051               1.
    CA082     RA093             3.44   RA094            -4.25
    CA082     RA095              -8.   RA096              24.
    CA083     R0000            .0075   RA019              -1.
    CA083     RA024             .068   RA026             .427
    CA083     RA028             .495   RA034              .01
    CA083     RA036           -.0029   RA057             .949
    CA084     RA019              -1.   RA051               1.
    CA084     RA093             5.64   RA094            -6.45
    CA084     RA095              8.5   RA096               4.
    CA085     R0000            .0075   RA020              -1.
    CA085     RA024             .113   RA026             .545
    CA085     RA029             .332   RA034              .01
    CA085     RA036           -.0032   RA057            1.074
    1CA085    R0000            .0075   1RA020             -1.
    1CA085    RA024             .109   RA026              .51
    1CA085    1RA029            .371   RA034              .01
    1CA085    RA036           -.0032   RA057            1.008
    CA086     RA020              -1.   RA048               1.
    CA086     RA089             -.56   RA090             -.25
    CA086     RA091               .5   RA092              22.
    1CA086    1RA020             -1.   RA048               1.
    1CA086    RA089             1.34   RA090            -2.15
    1CA086    RA091              3.5   RA092               8.
    CA087     RA020              -1.   RA051               1.
    CA087     RA093             -.56   RA094             -.25
    CA087     RA095              -8.   RA096              31.
    1CA087    1RA020             -1.   RA051               1.
    1CA087    RA093             1.34   RA094            -2.15
    1CA087    RA095              -5.   RA096              17.
    CA091     R0000            .0748   RA022              -1.
    CA091     RA039               1.   RA061            29.91
    CA091     RA062            -.494   RA063        -102.3101
    CA091     RA064        83.580099   RA065           10.074
    CA092     R0000            .0453   RA022              -1.
    CA092     RA040               1.   RA066             7.87
    CA092     RA067             -.67   RA068        -102.3101
    CA092     RA069        83.580099   RA070           10.074
    CA093     R0000            .0709   RA023              -1.
    CA093     RA039               1.   RA061            45.54
    CA093     RA062            -.468   RA063       -74.649899
    CA093     RA064             41.8   RA065           -1.642
    CA094     R0000            .0429   RA023              -1.
    CA094     RA040               1.   RA066            24.63
    CA094     RA067            -.635   RA068       -74.649899
    CA094     RA069             41.8   RA070           -1.642
    1CA095    R0000           -.0017   RA024              -1.
    1CA095    RA026               1.   RA032            -.034
    1CA095    RA034             .015   RA035             .019
    CA096     RA024              -1.   RA043               1.
    CA096     RA075             6.13   RA076            -7.58
    CA096     RA077             16.5   RA078            -3.52
    CA096     RA079              17.
    CA097     RA024              -1.   RA046               1.
    CA097     RA082             1.43   RA083            -5.26
    CA097     RA084            -3.52   RA085              6.5
    CA097     RA086              13.   RA087             -14.
    CA097     RA088               1.
    CA098     RA024              -1.   RA048               1.
    CA098     RA089           -12.56   RA090            11.75
    CA098     RA091             -3.5   RA092             -35.
    CA099     RA024              -1.   RA051               1.
    CA099     RA093           -12.56   RA094            11.75
    CA099     RA095             -12.   RA096             -26.
    CA100     R0000            .0534   RA026              -1.
    CA100     RA031             .255   RA032             .058
    CA100     RA033              .24   RA034             .082
    CA100     1RA034            .065   2RA034            .076
    CA100     RA035             .154   RA058             1.17
    CA100     1RA058              1.
    CA101     R0000            .0521   RA026              -1.
    CA101     RA031             .182   RA032             .059
    CA101     RA033              .24   RA034             .253
    CA101     1RA034             .07   2RA034            .076
    CA101     RA035             .065   RA058              .92
    CA102     R0000            .0067   RA031              .37
    CA102     1RA034             -1.   RA035              .48
    CA102     RA058              2.5
    1CA102    RA034               1.   1RA034             -1.
    CA103     R0000            .0155   RA031              .55
    CA103     RA032              .11   2RA034             -1.
    CA103     RA035              .17   RA058             2.83
    1CA103    RA034               1.   2RA034             -1.
    CA104     R0000             .036   RA019              -1.
    CA104     RA024             .077   RA026             .193
    CA104     RA028             .328   RA034             .008
    CA104     RA035             .001   RA036            -.006
    CA104     RA051             .276   RA054             .117
    CA104     RA093             2.88   RA094             -3.1
    CA104     RA095             6.35   RA096              1.1
    CA106     RA027              -1.   RA048               1.
    CA106     RA089            14.23   RA090           -15.04
    CA106     RA091              8.7   RA092             -34.
    1CA106    RA026            -.154   RA027            -.846
    1CA106    RA036            .0002   RA048               1.
    1CA106    RA057            -.078   RA089            12.39
    1CA106    RA090            -13.2   RA091               7.
    1CA106    RA092             -21.
    CA107     RA027              -1.   RA051               1.
    CA107     RA093            14.23   RA094           -15.04
    CA107     RA095               .2   RA096             -25.
    1CA107    RA026            -.154   RA027            -.846
    1CA107    RA036            .0002   RA051               1.
    1CA107    RA057            -.078   RA093            12.39
    1CA107    RA094            -13.2   RA095             -1.5
    1CA107    RA096             -12.
    CA108     RA028              -1.   RA051               1.
    CA108     RA093            14.47   RA094           -15.28
    CA108     RA095             19.5   RA096             -31.
    1CA108    RA026             -.13   RA028             -.87
    1CA108    RA036            .0002   RA051               1.
    1CA108    RA057            -.132   RA093            12.73
    1CA108    RA094           -13.54   RA095             17.2
    1CA108    RA096             -17.
    CA109     RA029              -1.   RA048               1.
    CA109     RA089             9.13   RA090            -9.94
    CA109     RA091              6.8   RA092             -38.
    1CA109    RA026            -.186   RA029            -.814
    1CA109    RA036            .0004   RA048               1.
    1CA109    RA057             -.05   RA089              7.8
    1CA109    RA090            -8.61   RA091              5.2
    1CA109    RA092             -28.
    2CA109    1RA029             -1.   RA048               1.
    2CA109    RA089             9.04   RA090            -9.85
    2CA109    RA091             10.5   RA092             -40.
    3CA109    RA026            -.176   1RA029           -.824
    3CA109    RA036            .0004   RA048               1.
    3CA109    RA057             -.09   RA089             7.94
    3CA109    RA090            -8.75   RA091              9.3
    3CA109    RA092             -30.
    CA110     RA029              -1.   RA051               1.
    CA110     RA093             9.13   RA094            -9.94
    CA110     RA095             -1.7   RA096             -29.
    1CA110    RA026            -.186   RA029            -.814
    1CA110    RA036            .0004   RA051               1.
    1CA110    RA057             -.05   RA093              7.8
    1CA110    RA094            -8.61   RA095             -3.3
    1CA110    RA096             -19.
    2CA110    1RA029             -1.   RA051               1.
    2CA110    RA093             9.04   RA094            -9.85
    2CA110    RA095               2.   RA096             -31.
    3CA110    RA026            -.176   1RA029           -.824
    3CA110    RA036            .0004   RA051               1.
    3CA110    RA057             -.09   RA093             7.94
    3CA110    RA094            -8.75   RA095               .8
    3CA110    RA096             -21.
    CA113     R0000            .0674   RA031              -1.
    CA113     RA039               1.   RA061             -.28
    CA113     RA062            -.445   RA063           -34.03
    CA113     RA064              18.   RA065            4.112
    CA114     R0000            .0407   RA031              -1.
    CA114     RA040               1.   RA066           -20.14
    CA114     RA067            -.604   RA068          -34.032
    CA114     RA069              18.   RA070            4.112
    CA115     R0000            .0576   RA032              -1.
    CA115     RA039               1.   RA061             -.24
    CA115     RA062             -.38   RA063        90.899899
    CA115     RA064            -32.8   RA065           -9.938
    1CA115    R0000            .0601   RA032              -1.
    1CA115    RA033              .25   RA039              .75
    1CA115    RA061             -.19   RA062            -.298
    1CA115    RA063             42.7   RA064            -24.5
    1CA115    RA065            -8.16
    CA116     R0000            .0348   RA032              -1.
    CA116     RA040               1.   RA066           -17.21
    CA116     RA067            -.516   RA068        90.899899
    CA116     RA069            -32.8   RA070           -9.938
    1CA116    R0000            .0363   RA032              -1.
    1CA116    RA033              .25   RA040              .75
    1CA116    RA066           -13.48   RA067            -.404
    1CA116    RA068             42.7   RA069            -24.5
    1CA116    RA070            -8.16
    CA117     R0000            .0218   RA033              -1.
    CA117     RA036           -.0409   RA044              .97
    CA117     RA059             1.97   1RA075            6.93
    CA117     1RA076            -8.3   1RA078           -2.96
    CA117     1RA079            3.88   1RA081            100.
    CA118     RA033              -1.   RA043               1.
    CA118     RA075             9.26   RA076           -10.71
    CA118     RA077             14.5   RA078            -3.05
    CA118     RA079               4.   RA081             100.
    CA120     RA033              -1.   RA046               1.
    CA120     RA082             4.57   RA083             -8.4
    CA120     RA084            -3.05   RA085              4.5
    CA120     RA087            -13.7   RA088              22.
    CA121     RA033              -1.   RA048               1.
    CA121     RA089           -18.06   RA090            17.25
    CA121     RA091             -5.5   RA092             -48.
    CA122     RA033              -1.   RA051               1.
    CA122     RA093           -18.06   RA094            17.25
    CA122     RA095             -14.   RA096             -39.
    CA123     RA046               1.   RA048              -1.
    CA123     RA082             9.81   RA083           -13.64
    CA123     RA085              10.   RA086              -2.
    CA123     RA087              71.   RA088              16.
    CA124     RA035              -1.   RA037               1.
    CA125     RA035              -1.   RA036            1.175
    CA126     RA036               1.   RA052              -1.
    CA127     RA043               1.   RA044              -1.
    CA127     RA077             -3.7
    CA128     R0000           4.1767   RA044              -1.
    CA128     RA045               1.   1RA081            -15.
    CA129     RA044              -1.   RA046               1.
    CA129     RA082            -4.69   RA083             2.31
    CA129     RA085            -13.5   RA086              -4.
    CA129     RA087             -14.   RA088             -14.
    CA130     RA046              -1.   RA047               1.
    CA130     RA082             -1.3   RA086              -6.
    CA130     RA087              13.   RA088              11.
    1CA130    RA045             -.75   RA047             -.25
    1CA130    1RA047              1.
    CA131     RA013             -.06   RA016             -.13
    CA131     RA018             -.81   1RA053              1.
    CA133     RA051              -1.   RA052               1.
    CA133     RA093            -2.94   RA094             1.75
    CA133     RA095              -6.   RA096             -11.
    2CA133    RA051              -1.   1RA052              1.
    2CA133    RA093            -2.26   RA094             2.01
    2CA133    RA095              -6.   RA096             -11.
    1CA134    RA013             -.25   RA018             -.75
    1CA134    RA049               1.
    2CA134    RA006             -.74   RA013             -.26
    2CA134    RA049               1.
    CA135     RA008              -1.   RA010               1.
    CA136     RA008              -1.   RA013               1.
    CA137     RA011               1.   RA012              -1.
    CA138     RA026              -1.   RA051               1.
    CA138     RA093            -3.13   RA094             2.32
    CA138     RA095              -6.   RA096              67.
    CA139     R0000              10.   RA018              -1.
    CA139     RA024             .115   RA026             .194
    CA139     RA027             .401   RA034              .04
    CA139     RA035             .001   RA036            -.022
    CA139     RA051             .129   RA054              .12
    CA139     RA093             .378   RA094            -.482
    CA139     RA095            -.194   RA096              12.
    CA140     R0000            .0431   RA020              -1.
    CA140     RA024             .069   RA026             .273
    CA140     RA034             .013   RA035             .001
    CA140     RA036             -.01   RA051               .4
    CA140     RA054             .244   RA093             3.38
    CA140     RA094             -3.7   RA095              2.8
    CA140     RA096              1.6
    CA142     RA027              -1.   RA038              .98
    CA143     RA007              -1.   RA038              .98
    CA144     R0000            .0075   RA006              -1.
    CA144     RA007             .488   RA024             .075
    CA144     RA026             .427   RA034              .01
    CA144     RA036           -.0029   RA057             .944
    CA145     RA006              -1.   RA048               1.
    CA145     RA089             5.14   RA090            -5.95
    CA145     RA091               .9   RA092               3.
    CA146     RA006              -1.   RA051               1.
    CA146     RA093             5.14   RA094            -5.95
    CA146     RA095             -7.6   RA096              12.
    CA147     RA007              -1.   RA048               1.
    CA147     RA089            15.26   RA090           -16.07
    CA147     RA091              8.9   RA092             -27.
    1CA147    RA007            -.897   RA026            -.103
    1CA147    RA036            .0002   RA048               1.
    1CA147    RA057            -.165   RA089            13.86
    1CA147    RA090           -14.67   RA091              7.7
    1CA147    RA092              -4.
    CA148     RA007              -1.   RA051               1.
    CA148     RA093            15.26   RA094           -16.07
    CA148     RA095               .4   RA096             -18.
    1CA148    RA007            -.897   RA026            -.103
    1CA148    RA036            .0002   RA051               1.
    1CA148    RA057            -.165   RA093            13.86
    1CA148    RA094           -14.67   RA095              -.8
    1CA148    RA096               5.
    CA154     R0000              .01   2RA005           1.008
    CA154     RA009             .241   RA011             .056
    CA154     RA012             .028   RA013             .065
    CA154     RA015             .077   RA017             .051
    CA154     RA019             .464   RA035             .018
    CA154     RA036           -.0316   RA055             1.17
    CA154     1RA055            1.17
    CA155     RA034              -1.   RA048               1.
    CA155     RA089           -11.76   RA090            10.95
    CA155     RA091             -4.5   RA092              -5.
    CA156     RA034              -1.   RA051               1.
    CA156     RA093           -11.76   RA094            10.95
    CA156     RA095             -13.   RA096               4.
    CA157     RA034              -1.   RA036               1.
    CA158     RA039              -1.   RA064           -2.684
    CA158     RA065            -1.61   RA097               1.
    CA166     RA040              -1.   RA066              5.4
    CA166     RA069           -1.379   RA070           -1.655
    CA166     1RA099              1.
    CA168     RA028              -1.   RA038              .98
    CB001     RB001             1.01   1RB005            .004
    CB001     RB006              .01   RB007             .028
    CB001     RB008             .135   RB010             .085
    CB001     RB012              .11   RB013              .09
    CB001     RB022              .23   RB031            -.038
    CB001     1RB031            .003   RB051               .3
    CB001     RB057               1.   RB091             .405
    CB001     RB092           -1.056   1RB092            -1.8
    CB001     RB093            -.174   1RB093           -4.95
    CB001     RB101             .969
    2CB001    RB001             1.01   1RB005            .004
    2CB001    RB006              .01   RB007             .028
    2CB001    RB008             .135   RB010             .085
    2CB001    RB012              .11   RB013              .09
    2CB001    RB022              .23   RB031            -.038
    2CB001    1RB031            .003   1RB053              .3
    2CB001    RB057               1.   RB101             .969
    2CB001    RB102             .633   RB103            -1.16
    2CB001    RB104             -4.8   RB105            -.174
    2CB001    RB106             -.45
    CB002     RB004             1.01   RB006             .003
    CB002     2RB007            .075   2RB008            .076
    CB002     RB010             .095   RB012              .14
    CB002     RB015              .13   RB022             .169
    CB002     RB025             .311   RB031            -.021
    CB002     1RB031            .001   RB058               1.
    CB002     1RB058              1.   RB099             .075
    CB002     RB101            1.012   UKLDF            -.075
    1CB002    RB001             1.01   1RB005            .004
    1CB002    RB006              .01   RB007             .018
    1CB002    RB008             .145   RB009             .044
    1CB002    RB011             .151   RB013              .09
    1CB002    RB016             .035   RB018               .5
    1CB002    RB031            -.026   1RB031            .003
    1CB002    RB058               1.   1RB058              1.
    1CB002    RB101             .969
    CB003     RB004             1.01   RB006             .003
    CB003     1RB007            .011   3RB008             .14
    CB003     RB010             .095   RB012              .14
    CB003     RB015              .13   RB022             .166
    CB003     RB025             .311   RB031            -.054
    CB003     1RB031            .001   RB057               1.
    CB003     RB101            1.012
    1CB003    RB004             1.01   RB006             .003
    1CB003    1RB007            .011   3RB008             .14
    1CB003    RB010             .095   RB012              .14
    1CB003    RB015              .13   RB022             .169
    1CB003    RB025             .311   RB031            -.021
    1CB003    1RB031            .001   RB058               1.
    1CB003    1RB058              1.   RB100             -.56
    1CB003    RB101            1.012
    2CB003    RB003             1.01   1RB005            .002
    2CB003    RB006             .012   RB007             .056
    2CB003    RB008             .089   RB010             .076
    2CB003    RB012              .09   1RB013              .1
    2CB003    RB016             .032   1RB020            .541
    2CB003    RB031            -.023   1RB031            .002
    2CB003    RB058               1.   1RB058              1.
    2CB003    RB101            1.021
    1CB004    1RB001            1.01   RB006             .008
    1CB004    2RB007              .1   2RB008             .09
    1CB004    RB009             .047   RB011             .161
    1CB004    1RB013             .09   1RB016             .04
    1CB004    1RB018            .463   RB031            -.025
    1CB004    1RB031            .001   RB058               1.
    1CB004    RB099               .1   RB101             .969
    1CB004    UKLDF              -.1
    2CB004    1RB001            1.01   RB006             .008
    2CB004    1RB007            .047   1RB008            .143
    2CB004    RB009             .047   RB011             .161
    2CB004    1RB013             .09   1RB016             .04
    2CB004    1RB018            .463   RB031           -.0251
    2CB004    1RB031            .001   RB058               1.
    2CB004    RB101             .969
    1CB005    RB002             1.01   1RB005            .004
    1CB005    RB006             .011   1RB007             .04
    1CB005    1RB008            .102   RB009             .035
    1CB005    RB011             .121   1RB013            .083
    1CB005    1RB016             .04   RB019              .56
    1CB005    RB031            -.023   1RB031            .004
    1CB005    RB058               1.   RB101               1.
    2CB005    RB002             1.01   1RB005            .004
    2CB005    RB006             .011   2RB007            .077
    2CB005    2RB008            .065   RB009             .035
    2CB005    RB011             .121   1RB013            .083
    2CB005    1RB016             .04   RB019              .56
    2CB005    RB031            -.023   1RB031            .004
    2CB005    RB058               1.   RB099             .077
    2CB005    RB101               1.   UKLDF            -.077
    CB006     RB005             1.01   1RB005            .001
    CB006     RB006             .004   1RB007            .024
    CB006     RB008             .073   RB010             .059
    CB006     RB012             .083   RB014             .096
    CB006     RB020             .509   RB022             .146
    CB006     RB031            -.038   1RB031            .001
    CB006     RB057               1.   RB101             1.05
    1CB006    RB005             1.01   1RB005            .001
    1CB006    RB006             .004   1RB007            .024
    1CB006    RB008             .073   RB010             .059
    1CB006    RB012             .083   RB014             .096
    1CB006    RB020             .659   RB031            -.021
    1CB006    1RB031            .001   RB058               1.
    1CB006    1RB058              1.   RB101             1.05
    2CB006    1RB005             -1.   2RB005              1.
    3CB006    RB005             1.01   1RB005            .001
    3CB006    RB006             .004   2RB007            .071
    3CB006    2RB008            .026   RB010             .059
    3CB006    RB012             .083   RB014             .096
    3CB006    RB020             .659   RB031            -.021
    3CB006    1RB031            .001   RB058               1.
    3CB006    1RB058              1.   RB099             .071
    3CB006    RB101             1.05   UKLDF            -.071
    CB007     1RB004            1.01   RB006             .005
    CB007     1RB007            .025   3RB008            .058
    CB007     RB010              .06   RB012              .07
    CB007     1RB015            .311   1RB025             .47
    CB007     RB031            -.026   1RB031            .001
    CB007     RB058               1.   1RB058              1.
    CB007     RB101            1.012
    1CB007    1RB004            1.01   RB006             .005
    1CB007    2RB007             .06   2RB008            .023
    1CB007    RB010              .06   RB012              .07
    1CB007    1RB015            .311   1RB025             .47
    1CB007    RB031            -.026   1RB031            .001
    1CB007    RB058               1.   1RB058              1.
    1CB007    RB099              .06   RB101            1.012
    1CB007    UKLDF             -.06
    2CB007    R0000             100.   1RB005              1.
    2CB008    R0000              7.5   2RB005              1.
    2CB009    RB006              -1.   1RB031              1.
    2CB010    1RB006             -1.   1RB031              1.
    CB011     RB009            -.227   RB010             .461
    CB011     RB011            -.773   RB012             .539
    2CB012    R0000               7.   1RB006              1.
    1CB013    1RB007             -1.   2RB007              1.
    1CB013    RB099               1.   UKLDF              -1.
    2CB013    RB031             1.16   1RB031             -1.
    3CB013    R0000             .221   RB007              -1.
    3CB013    RB031            -.067   RB034              .96
    3CB013    RB063            1.143   RB064            30.51
    3CB013    RB065           -70.73   RB066            45.47
    3CB013    RB067            -.404
    1CB014    RB007               1.   1RB007             -1.
    2CB014    RB009            -.046   RB010            -.622
    2CB014    RB011            -.288   RB013             -.08
    2CB014    2RB040              1.
    3CB014    R0000             .097   RB007              -1.
    3CB014    RB031            -.067   RB036              .96
    3CB014    RB063            1.143   RB068            17.03
    3CB014    RB069           -70.73   RB070            45.47
    3CB014    RB071            -.647
    CB015     RB007              -1.   RB038               1.
    CB015     UKLDF              -1.
    1CB015    RB007              -.5   2RB008             -.5
    1CB015    RB038               1.   UKLDF              -1.
    2CB015    RB009            -.046   RB010            -.622
    2CB015    RB011            -.288   1RB013            -.08
    2CB015    2RB040              1.
    1CB016    RB007              -1.   RB031            -.067
    1CB016    RB039              .96   RB063            1.143
    1CB016    RB072             16.7   RB073           -10.46
    1CB016    RB074            -33.6
    2CB016    2RB005             -1.   1RB031              1.
    CB017     RB007              -1.   RB031             1.11
    2CB017    2RB005             -.1   RB006              -.9
    2CB017    1RB006              1.
    CB018     R0000             .087   3RB008             -1.
    CB018     RB036              .99   RB068            14.05
    CB018     RB069             23.7   RB070            -31.1
    CB018     RB071            -.588
    1CB018    RB007              -.4   3RB008             -.6
    1CB018    RB038               1.   UKLDF              -1.
    CB019     RB007              -.4   RB008              -.6
    CB019     RB038               1.   UKLDF              -1.
    1CB019    RB007              -.4   1RB008             -.6
    1CB019    RB038               1.   UKLDF              -1.
    CB020     3RB008             -1.   RB039              .99
    CB020     RB072            -.396   RB073              .54
    CB020     RB074           -34.65
    1CB020    RB008              -1.   RB031            -.067
    1CB020    RB039              .98   RB063            1.667
    1CB020    RB072             2.06   RB073             .539
    1CB020    RB074            -34.3
    2CB020    1RB008             -1.   RB031            -.067
    2CB020    RB039              .98   RB063               1.
    2CB020    RB072             3.92   RB073             .539
    2CB020    RB074            -34.3
    CB021     R0000             .088   RB008              -1.
    CB021     RB031            -.067   RB036              .98
    CB021     RB063            1.667   RB068            21.58
    CB021     RB069            14.56   RB070           -30.46
    CB021     RB071            -.593
    1CB021    R0000             .089   1RB008             -1.
    1CB021    RB031            -.067   RB036              .98
    1CB021    RB063               1.   RB068            21.88
    1CB021    RB069            14.76   RB070           -30.87
    1CB021    RB071            -.601
    2CB021    RB030              -1.   2RB030              1.
    2CB022    1RB030             -1.   2RB030              1.
    CB029     RB008              -1.   RB031             1.11
    1CB029    1RB008             -1.   RB031             1.11
    2CB029    2RB008             -1.   RB031              1.1
    2CB030    R0000             .056   2RB005            .036
    2CB030    RB006             .052   RB007             .029
    2CB030    RB008              -1.   RB024             .765
    2CB030    RB031             .027   1RB031            .009
    2CB030    RB060               1.
    2CB031    R0000             .094   2RB005            .044
    2CB031    RB006             .075   RB007             .029
    2CB031    RB008              -1.   1RB024            .714
    2CB031    RB031             .027   1RB031            .024
    2CB031    RB060               1.
    1CB032    R0000             .058   2RB005            .017
    1CB032    RB006             .024   RB007             .029
    1CB032    3RB008             -1.   RB024             .861
    1CB032    RB031             .027   1RB031            -.04
    1CB032    RB060            1.712
    2CB032    R0000              .06   2RB005            .046
    2CB032    RB006             .077   1RB007            .029
    2CB032    1RB008             -1.   RB024             .706
    2CB032    RB031             .027   1RB031            .033
    2CB032    RB060            1.022
    1CB033    R0000             .096   2RB005            .026
    1CB033    RB006             .044   RB007             .029
    1CB033    3RB008             -1.   1RB024             .81
    1CB033    RB031             .027   1RB031           -.023
    1CB033    RB060            1.712
    2CB033    R0000               .1   2RB005            .055
    2CB033    RB006             .101   1RB007            .029
    2CB033    1RB008             -1.   1RB024             .65
    2CB033    RB031             .027   1RB031            .051
    2CB033    RB060            1.022
    CB034     RB009              -1.   RB039              .97
    CB034     RB072            -3.01   RB073             1.84
    CB034     RB074           -33.95
    1CB034    R0000             .213   2RB005            .034
    1CB034    RB006            .0415   1RB007             .01
    1CB034    2RB008             -1.   RB031             .037
    1CB034    1RB031            .004   RB034            .7925
    1CB034    RB060            .9915   RB064              3.7
    1CB034    RB065            7.773   RB066          -10.527
    1CB034    RB067            -.287
    2CB034    R0000              .22   2RB005            .028
    2CB034    RB006             .022   RB007              .01
    2CB034    RB008            -.766   RB009            -.234
    2CB034    RB031             .037   1RB031           -.012
    2CB034    RB034             .834   RB060             .983
    2CB034    RB064             3.86   RB065            9.648
    2CB034    RB066          -11.888   RB067              -.3
    1CB035    R0000            .0585   2RB005            .034
    1CB035    RB006            .0415   1RB007             .01
    1CB035    2RB008             -1.   RB026             .355
    1CB035    RB027            .4375   RB031            .0255
    1CB035    1RB031            .004   RB060            .9915
    2CB035    R0000             .058   2RB005            .028
    2CB035    RB006             .022   RB007              .01
    2CB035    RB008            -.766   RB009            -.234
    2CB035    RB026             .374   RB027              .46
    2CB035    RB031             .025   1RB031           -.012
    2CB035    RB060             .983
    1CB036    R0000             .239   2RB005           .0435
    1CB036    RB006             .064   1RB007             .01
    1CB036    2RB008             -1.   RB031             .037
    1CB036    1RB031           .0225   RB034             .737
    1CB036    RB060            .9915   RB064           -.2815
    1CB036    RB065            5.696   RB066           -8.554
    1CB036    RB067            -.263
    2CB036    R0000             .248   2RB005            .035
    2CB036    RB006              .04   RB007              .01
    2CB036    RB008            -.766   RB009            -.234
    2CB036    RB031             .037   1RB031            .001
    2CB036    RB034             .791   RB060             .983
    2CB036    RB064              -.3   RB065            7.007
    2CB036    RB066          -10.218   RB067             -.28
    1CB037    R0000             .098   2RB005           .0435
    1CB037    RB006             .064   1RB007             .01
    1CB037    2RB008             -1.   RB026            .2375
    1CB037    RB027            .4995   RB031             .026
    1CB037    1RB031           .0225   RB060            .9915
    2CB037    R0000             .097   2RB005            .035
    2CB037    RB006              .04   RB007              .01
    2CB037    RB008            -.766   RB009            -.234
    2CB037    RB026             .255   RB027             .536
    2CB037    RB031             .025   1RB031            .001
    2CB037    RB060             .983
    2CB038    R0000             .207   2RB005             .04
    2CB038    RB006             .061   1RB007             .01
    2CB038    1RB008           -.676   RB009            -.324
    2CB038    RB031             .037   1RB031             .02
    2CB038    RB034             .751   RB060               1.
    2CB038    RB064             3.54   RB065            5.898
    2CB038    RB066           -9.167   RB067            -.275
    CB039     RB010              -1.   RB039              .97
    CB039     RB072            -5.43   RB073             1.73
    CB039     RB074           -33.95
    1CB039    RB010              -1.   RB047               1.
    1CB039    RB081               3.   RB082            -24.5
    1CB039    RB083            -9.22   RB084               .3
    1CB039    RB085           23.573   RB086            11.42
    2CB039    R0000             .059   2RB005             .04
    2CB039    RB006             .061   1RB007             .01
    2CB039    1RB008           -.676   RB009            -.324
    2CB039    RB026             .336   RB027             .415
    2CB039    RB031             .026   1RB031             .02
    2CB039    RB060               1.
    CB040     RB010              -1.   RB012             .034
    CB040     RB031             .025   RB040              .88
    1CB040    RB010              -1.   RB045               1.
    1CB040    RB075              2.2   RB076              -9.
    1CB040    RB077            -9.22   RB078               .3
    1CB040    RB079           23.573   RB080             -10.
    2CB040    R0000             .231   2RB005            .052
    2CB040    RB006             .088   1RB007             .01
    2CB040    1RB008           -.676   RB009            -.324
    2CB040    RB031             .037   1RB031            .044
    2CB040    RB034             .683   RB060               1.
    2CB040    RB064            -.263   RB065            4.385
    2CB040    RB066            -6.89   RB067            -.246
    CB041     RB008             .049   RB009            -.194
    CB041     RB010             -.15   RB011            -.704
    CB041     RB042             .965
    2CB041    R0000             .099   2RB005            .052
    2CB041    RB006             .088   1RB007             .01
    2CB041    1RB008           -.676   RB009            -.324
    2CB041    RB026              .22   RB027             .463
    2CB041    RB031             .027   1RB031            .044
    2CB041    RB060               1.
    CB043     RB009            -.227   RB011            -.773
    CB043     RB012               .2   RB031            -.027
    CB043     RB041             .772   RB062               1.
    CB044     RB009            -.227   RB011            -.773
    CB044     RB045               1.   RB075              3.3
    CB044     RB076             -9.1   RB077             -7.2
    CB044     RB078             -1.2   RB079             6.36
    CB044     RB080             -10.
    CB045     RB009            -.227   RB011            -.773
    CB045     RB047               1.   RB081              4.1
    CB045     RB082            -24.6   RB083             -7.2
    CB045     RB084             -1.2   RB085             6.36
    CB045     RB086              8.8
    CB046     RB009            -.227   RB011            -.773
    CB046     RB046               1.   1RB050              1.
    CB046     RB087            -22.9   1RB087            21.4
    CB046     RB088            -29.7   RB089            5.097
    CB046     RB090             -25.   1RB090            -31.
    CB047     RB009            -.227   RB011            -.773
    CB047     RB051               1.   RB091           -30.43
    CB047     RB092            29.26   1RB092            -67.
    CB047     RB093            5.097   1RB093           -37.7
    1CB047    RB009            -.227   RB011            -.773
    1CB047    1RB053              1.   RB102           -29.67
    1CB047    RB103            27.92   RB104             -77.
    1CB047    RB105            5.097   RB106            -22.7
    CB048     RB009            -.227   RB011            -.773
    CB048     RB053               1.   RB094            -31.3
    CB048     RB095            30.59   RB096            -17.7
    CB048     RB097             -83.
    CB049     RB011              -1.   RB039              .97
    CB049     RB072            -8.54   RB073             1.55
    CB049     RB074            63.05
    CB051     RB011              -1.   RB012               .2
    CB051     RB031            -.027   RB041             .772
    CB051     RB062               1.
    CB052     RB011              -1.   RB045               1.
    CB052     RB075              2.5   RB076             -9.1
    CB052     RB077             -6.9   RB078             -1.2
    CB052     RB079            2.069   RB080             -10.
    CB053     RB011              -1.   RB047               1.
    CB053     RB081              3.3   RB082            -24.5
    CB053     RB083             -6.9   RB084             -1.2
    CB053     RB085            2.069   RB086              8.8
    CB054     RB011              -1.   RB046               1.
    CB054     1RB050              1.   RB087           -22.69
    CB054     1RB087           21.19   RB088            -29.4
    CB054     RB089            2.069   RB090             -25.
    CB054     1RB090            -31.
    CB055     RB011              -1.   RB051               1.
    CB055     RB091           -30.22   RB092            29.05
    CB055     1RB092            -67.   RB093            2.069
    CB055     1RB093           -37.4
    1CB055    RB011              -1.   1RB053              1.
    1CB055    RB102           -29.46   RB103            27.71
    1CB055    RB104             -77.   RB105            2.069
    1CB055    RB106            -22.4
    CB056     RB011              -1.   RB053               1.
    CB056     RB094            -29.9   RB095            29.19
    CB056     RB096            -17.4   RB097             -83.
    1CB057    RB012            -.865   RB031            -.027
    1CB057    RB041             .386   1RB043            .386
    1CB057    RB062               2.
    2CB057    RB010              -1.   RB012             .216
    2CB057    RB031            -.012   RB040             .364
    2CB057    1RB040            .364   RB062             1.82
    CB058     RB012              -1.   RB045               1.
    CB058     RB075               1.   RB076              -9.
    CB058     RB077             -6.1   RB078             -1.2
    CB058     RB079             1.52   RB080             -10.
    CB059     RB012              -1.   RB047               1.
    CB059     RB081              1.8   RB082            -24.1
    CB059     RB083             -6.1   RB084             -1.2
    CB059     RB085             1.52   RB086             7.42
    CB060     RB012              -1.   RB046               1.
    CB060     1RB050              1.   RB087           -22.69
    CB060     1RB087           21.19   RB088            -28.6
    CB060     RB089             1.52   RB090             -25.
    CB060     1RB090            -31.
    CB061     RB012              -1.   RB051               1.
    CB061     RB091           -30.22   RB092            29.05
    CB061     1RB092            -67.   RB093             1.52
    CB061     1RB093           -36.6
    1CB061    RB012              -1.   1RB053              1.
    1CB061    RB102           -29.46   RB103            27.71
    1CB061    RB104             -77.   RB105             1.52
    1CB061    RB106            -21.6
    CB062     RB012              -1.   RB053               1.
    CB062     RB094            -27.9   RB095            27.19
    CB062     RB096            -16.6   RB097             -83.
    CB063     RB013              -1.   RB045               1.
    CB063     RB075             -3.4   RB076             12.9
    CB063     RB077             -1.5   RB078             -.25
    CB063     RB079            -4.15   RB080             -10.
    1CB063    1RB013             -1.   RB045               1.
    1CB063    RB075             -2.7   RB076              7.6
    1CB063    RB077              2.4   RB078             -.25
    1CB063    RB079            -4.07   RB080             -10.
    CB064     RB013              -1.   RB047               1.
    CB064     RB081             -2.6   RB082             -1.7
    CB064     RB083             -1.5   RB084             -.25
    CB064     RB085            -4.15   RB086             2.16
    1CB064    1RB013             -1.   RB047               1.
    1CB064    RB081             -1.9   RB082             -7.1
    1CB064    RB083              2.4   RB084             -.25
    1CB064    RB085            -4.07   RB086             2.16
    CB065     RB013              -1.   RB046               1.
    CB065     1RB050              1.   RB087            -14.9
    CB065     1RB087            13.4   RB088             -26.
    CB065     RB089            -4.41   RB090             -25.
    CB065     1RB090            -22.
    1CB065    1RB013             -1.   RB046               1.
    1CB065    1RB050              1.   RB087            -14.9
    1CB065    1RB087            13.4   RB088            -23.3
    1CB065    RB089            -4.41   RB090             -25.
    1CB065    1RB090            -22.
    CB066     RB014              -1.   RB045               1.
    CB066     RB075             -6.1   RB076              19.
    CB066     RB077              -1.   RB078               .2
    CB066     RB079           -3.959   RB080             -10.
    CB067     RB014              -1.   RB047               1.
    CB067     RB081             -5.3   RB082              4.6
    CB067     RB083              -1.   RB084               .2
    CB067     RB085           -3.959   RB086            -2.33
    CB068     RB014              -1.   RB046               1.
    CB068     1RB050              1.   RB087           -12.03
    CB068     1RB087           10.53   RB088             -23.
    CB068     RB089           -3.959   RB090             -25.
    CB068     1RB090            -17.
    CB069     RB015              -1.   RB045               1.
    CB069     RB075             -5.3   RB076             14.4
    CB069     RB077             -8.3   RB078               .3
    CB069     RB079            -3.43   RB080             -10.
    1CB069    RB012            -.174   RB013             -.16
    1CB069    RB014            -.186   RB015              -.4
    1CB069    RB016             -.08   RB031            -.025
    1CB069    1RB045              1.
    3CB069    RB012             -.25   1RB013            -.15
    3CB069    RB014             -.15   RB015              -.4
    3CB069    1RB016            -.05   RB031            -.025
    3CB069    1RB045              1.
    CB070     RB015              -1.   RB047               1.
    CB070     RB081             -4.5   RB083             -8.3
    CB070     RB084               .3   RB085            -3.43
    CB070     RB086            -2.05
    1CB070    1RB015             -1.   RB045               1.
    1CB070    RB075             -8.9   RB076              -.6
    1CB070    RB077             -7.8   RB078              1.6
    1CB070    RB079            -3.43   RB080             -10.
    2CB070    1RB015             -1.   RB047               1.
    2CB070    RB081             -8.1   RB082            -14.6
    2CB070    RB083             -7.8   RB084              1.6
    2CB070    RB085            -3.43   RB086            -1.76
    CB071     RB015              -1.   RB046               1.
    CB071     1RB050              1.   RB087           -12.24
    CB071     1RB087           10.74   RB088            -29.9
    CB071     RB089           -3.526   RB090             -25.
    CB071     1RB090            -18.
    1CB071    2RB015             -1.   RB045               1.
    1CB071    RB075               .1   RB076             19.4
    1CB071    RB077             -8.7   RB078             -1.8
    1CB071    RB079            -4.41   RB080              -5.
    2CB071    2RB015             -1.   RB047               1.
    2CB071    RB081               .9   RB082              5.5
    2CB071    RB083             -8.7   RB084             -1.8
    2CB071    RB085            -4.41   RB086               .7
    CB072     RB015              -1.   RB051               1.
    CB072     RB091           -19.77   RB092             18.6
    CB072     1RB092            -54.   RB093           -3.526
    CB072     1RB093          -39.06
    1CB072    RB015              -1.   1RB053              1.
    1CB072    RB102           -19.01   RB103            17.26
    1CB072    RB104             -64.   RB105           -3.526
    1CB072    RB106           -24.06
    CB073     RB015              -1.   RB053               1.
    CB073     RB094           -20.57   RB095            19.86
    CB073     RB096            -19.1   RB097             -70.
    CB074     RB016              -1.   RB047               1.
    CB074     RB081             -5.3   RB082            120.2
    CB074     RB083               4.   RB084              -.3
    CB074     RB085            -3.56   RB086            -2.05
    1CB074    1RB016             -1.   RB047               1.
    1CB074    RB081              -6.   RB082            119.5
    1CB074    RB083              9.8   RB084             -.45
    1CB074    RB085            -3.56   RB086             -2.6
    CB075     RB016              -1.   RB051               1.
    CB075     RB091           -19.77   RB092             18.6
    CB075     1RB092            -32.   RB093           -3.856
    CB075     1RB093            -29.
    1CB075    1RB016             -1.   RB051               1.
    1CB075    RB091           -19.19   RB092            18.02
    1CB075    1RB092            -32.   RB093           -3.856
    1CB075    1RB093            -24.
    2CB075    RB016              -1.   1RB053              1.
    2CB075    RB102           -19.01   RB103            17.26
    2CB075    RB104             -42.   RB105           -3.856
    2CB075    RB106             -14.
    3CB075    1RB016             -1.   1RB053              1.
    3CB075    RB102           -18.43   RB103            16.68
    3CB075    RB104             -42.   RB105           -3.856
    3CB075    RB106              -9.
    CB076     R0000             .095   2RB005            .028
    CB076     RB006              .08   RB016              -1.
    CB076     RB028             .126   RB029             .155
    CB076     RB030             .405   RB031            -.051
    CB076     1RB031            .087   RB051             .058
    CB076     RB061            1.467   RB091           -1.269
    CB076     RB092              1.2   1RB092             3.2
    CB076     RB093            -.164   1RB093          -1.288
    1CB076    R0000             .095   2RB005            .032
    1CB076    RB006             .068   1RB016             -1.
    1CB076    RB028             .126   RB029             .155
    1CB076    1RB030            .405   RB031            -.051
    1CB076    1RB031            .095   RB051             .058
    1CB076    RB061            1.467   RB091           -1.269
    1CB076    RB092              1.2   1RB092             3.2
    1CB076    RB093            -.164   1RB093          -1.288
    CB077     RB025              -1.   RB051               1.
    CB077     RB091            -2.61   RB092             1.44
    CB077     1RB092             82.   RB093            -.643
    CB077     1RB093           -35.8
    1CB077    RB025              -1.   1RB053              1.
    1CB077    RB102            -1.85   RB103               .1
    1CB077    RB104              72.   RB105            -.643
    1CB077    RB106            -20.8
    CB078     1RB025             -1.   RB053               1.
    CB078     RB095             -.71   RB096             -15.
    CB078     RB097             -10.
    1CB078    1RB025             -1.   RB051               1.
    1CB078    RB091             -.76   RB092             -.41
    1CB078    1RB092              6.   RB093            -.643
    1CB078    1RB093            -35.
    2CB078    2RB025             -1.   RB051               1.
    2CB078    RB091            -1.52   RB092              .35
    2CB078    1RB092            25.7   RB093            -.803
    2CB078    1RB093            -36.
    3CB078    2RB025             -1.   RB053               1.
    3CB078    RB094             -.76   RB095              .05
    3CB078    RB096             -16.   RB097              9.7
    4CB078    1RB025             -1.   1RB053              1.
    4CB078    RB103            -1.75   RB104              -4.
    4CB078    RB105            -.643   RB106             -20.
    CB079     1RB020             -1.   RB053               1.
    CB079     RB094             1.34   RB095            -2.05
    CB079     RB096              5.4   RB097             -15.
    1CB079    2RB025             -1.   1RB053              1.
    1CB079    RB102             -.76   RB103             -.99
    1CB079    RB104             15.7   RB105            -.803
    1CB079    RB106             -21.
    CB080     1RB020             -1.   RB051               1.
    CB080     RB091              .58   RB092            -1.75
    CB080     1RB092              1.   RB093            -1.93
    CB080     1RB093           -14.6
    1CB080    1RB020             -1.   1RB053              1.
    1CB080    RB102             1.34   RB103            -3.09
    1CB080    RB104              -9.   RB105            -1.93
    1CB080    RB106               .4
    CB081     RB018              -1.   RB051               1.
    CB081     RB091            -2.41   RB092             1.24
    CB081     1RB092              1.   RB093            -.803
    CB081     1RB093            -18.
    1CB081    1RB018             -1.   RB051               1.
    1CB081    RB091            -2.05   RB092              .88
    1CB081    1RB092             -7.   RB093            -.803
    1CB081    1RB093             -4.
    2CB081    RB019              -1.   RB051               1.
    2CB081    RB091             -.03   RB092            -1.14
    2CB081    1RB092            -11.   RB093            -.803
    2CB081    1RB093             -.7
    3CB081    RB018              -1.   1RB053              1.
    3CB081    RB102            -1.41   RB103             -.34
    3CB081    RB104              -9.   RB105            -.803
    3CB081    RB106              -3.
    4CB081    1RB018             -1.   1RB053              1.
    4CB081    RB102            -1.32   RB103             -.43
    4CB081    RB104             -14.   RB105            -.803
    4CB081    RB106              11.
    CB082     RB018              -1.   RB053               1.
    CB082     RB094            -1.65   RB095              .94
    CB082     RB096               2.   RB097             -15.
    1CB082    1RB018             -1.   RB053               1.
    1CB082    RB094            -1.29   RB095              .58
    1CB082    RB096              16.   RB097             -23.
    2CB082    RB019              -1.   RB053               1.
    2CB082    RB094              .73   RB095            -1.44
    2CB082    RB096             19.3   RB097             -27.
    3CB082    RB019              -1.   1RB053              1.
    3CB082    RB102              .73   RB103            -2.48
    3CB082    RB104             -21.   RB105            -.803
    3CB082    RB106             14.3
    1CB083    1RB018             -1.   1RB021            .059
    1CB083    1RB022            .358   1RB023             .42
    1CB083    1RB030           -.054   RB031            -.075
    1CB083    RB048             .135   RB051               .1
    1CB083    RB059            1.026   RB091            -.374
    1CB083    RB092             .162   1RB092             4.3
    1CB083    RB093            -.159   1RB093            .996
    2CB083    RB019              -1.   1RB021            .063
    2CB083    1RB022            .317   1RB023            .484
    2CB083    1RB030           -.063   RB031            -.074
    2CB083    RB048             .135   RB051             .082
    2CB083    RB059            1.025   RB091            -.317
    2CB083    RB092             .115   1RB092             3.5
    2CB083    RB093            -.156   1RB093           1.232
    1CB085    1RB018             -1.   1RB021            .057
    1CB085    1RB022            .308   1RB023            .396
    1CB085    1RB030            -.06   RB031            -.094
    1CB085    RB048             .189   RB051              .14
    1CB085    RB059            1.036   RB091            -.524
    1CB085    RB092             .361   1RB092              6.
    1CB085    RB093            -.223   1RB093             1.4
    2CB085    RB019              -1.   1RB021            .061
    2CB085    1RB022            .273   1RB023            .456
    2CB085    1RB030            -.07   RB031            -.094
    2CB085    RB048             .193   RB051             .117
    2CB085    RB059            1.036   RB091            -.453
    2CB085    RB092             .316   1RB092              5.
    2CB085    RB093            -.223   1RB093            1.76
    CB086     RB016              -1.   RB045               1.
    CB086     RB075             -6.1   RB076         134.6001
    CB086     RB077               4.   RB078              -.3
    CB086     RB079            -3.56   RB080             -10.
    1CB086    1RB016             -1.   RB045               1.
    1CB086    RB075             -6.8   RB076       133.699999
    1CB086    RB077              9.8   RB078             -.45
    1CB086    RB079            -3.56   RB080             -10.
    CB089     RB018              -1.   RB046               1.
    CB089     1RB050              1.   RB087              5.4
    CB089     1RB087            -6.9   RB088             -10.
    CB089     RB089            -.803   RB090              75.
    CB089     1RB090             37.
    CB090     1RB018             -1.   RB046               1.
    CB090     1RB050              1.   RB087             5.59
    CB090     1RB087           -7.09   RB088               4.
    CB090     RB089            -.803   RB090              75.
    CB090     1RB090             29.
    1CB090    RB019              -1.   RB046               1.
    1CB090    1RB050              1.   RB087             7.64
    1CB090    1RB087           -9.04   RB088              7.3
    1CB090    RB089            -.803   RB090              75.
    1CB090    1RB090             25.
    CB091     RB011            -.018   RB020              -1.
    CB091     1RB021            .055   RB022             .359
    CB091     RB031            -.046   2RB049            .222
    CB091     RB051             .364   RB091             3.83
    CB091     RB092            -4.26   1RB092           -12.7
    CB091     RB093            -.234   1RB093           -4.11
    CB092     RB020              -1.   RB046               1.
    CB092     1RB050              1.   RB087             8.97
    CB092     1RB087          -10.47   RB088             -8.5
    CB092     RB089            -.803   RB090             -25.
    CB092     1RB090             15.
    CB093     RB020              -1.   RB051               1.
    CB093     RB091              1.3   RB092            -2.47
    CB093     1RB092            -21.   RB093            -.803
    CB093     1RB093           -16.5
    1CB093    RB020              -1.   1RB053              1.
    1CB093    RB102             2.06   RB103            -3.81
    1CB093    RB104             -31.   RB105            -.803
    1CB093    RB106             -1.5
    CB094     RB020              -1.   RB053               1.
    CB094     RB094             2.06   RB095            -2.77
    CB094     RB096              3.5   RB097             -37.
    2CB097    R0000             .095   2RB005            .021
    2CB097    RB006             .065   RB022              -1.
    2CB097    RB028             .139   RB029             .171
    2CB097    RB030             .278   RB031            -.036
    2CB097    1RB031             .06   RB051             .207
    2CB097    RB061               1.   RB091           -2.637
    2CB097    RB092            2.395   1RB092             4.1
    2CB097    RB093            -.406   1RB093           -4.93
    2CB098    R0000             .095   2RB005            .021
    2CB098    RB006             .065   RB022              -1.
    2CB098    RB028             .139   RB029             .171
    2CB098    RB030             .278   RB031             .102
    2CB098    1RB031             .06   RB051             .069
    2CB098    RB061               1.   RB091            -.574
    2CB098    RB092             .493   1RB092             3.9
    2CB098    1RB093           -1.53
    3CB098    1RB021             -1.   1RB053              1.
    3CB098    RB102           -17.62   RB103            15.87
    3CB098    RB104             -47.   RB105           -3.404
    3CB098    RB106             -5.2
    2CB099    R0000             .095   2RB005            .031
    2CB099    RB006             .082   RB022              -1.
    2CB099    RB028              .15   RB029             .185
    2CB099    RB030             .231   RB031            -.036
    2CB099    1RB031            .082   RB051             .167
    2CB099    RB061            1.412   RB091           -2.194
    2CB099    RB092            1.994   1RB092             3.3
    2CB099    RB093            -.356   1RB093           -3.99
    1CB100    R0000             .095   2RB005            .024
    1CB100    RB006             .059   1RB021             -1.
    1CB100    RB028             .125   RB029             .155
    1CB100    1RB030            .398   RB031            -.051
    1CB100    1RB031            .085   RB051               .1
    1CB100    RB061               1.   RB091           -2.188
    1CB100    RB092             2.07   1RB092             5.6
    1CB100    RB093            -.284   1RB093           -2.34
    2CB100    R0000             .095   2RB005            .031
    2CB100    RB006             .082   RB022              -1.
    2CB100    RB028              .15   RB029             .185
    2CB100    RB030             .231   RB031             .085
    2CB100    1RB031            .082   RB051             .046
    2CB100    RB061            1.412   RB091            -.382
    2CB100    RB092             .329   1RB092             2.6
    2CB100    1RB093           -1.02
    1CB101    R0000             .095   2RB005            .036
    1CB101    RB006             .075   1RB021             -1.
    1CB101    RB028             .136   RB029             .167
    1CB101    1RB030            .336   RB031            -.051
    1CB101    1RB031            .106   RB051             .077
    1CB101    RB061              1.1   RB091           -1.684
    1CB101    RB092             1.59   1RB092             4.3
    1CB101    RB093            -.218   1RB093            -1.8
    2CB101    R0000             .095   2RB005            .024
    2CB101    RB006             .055   1RB022             -1.
    2CB101    RB028             .139   RB029             .171
    2CB101    1RB030            .278   RB031            -.036
    2CB101    1RB031            .065   RB051             .207
    2CB101    RB061               1.   RB091           -2.636
    2CB101    RB092            2.395   1RB092             4.1
    2CB101    RB093            -.406   1RB093           -3.79
    CB102     RB022              -1.   RB051               1.
    CB102     RB091            -9.08   RB092             7.91
    CB102     1RB092             43.   RB093           -2.549
    CB102     1RB093            -18.
    1CB102    1RB022             -1.   RB051               1.
    1CB102    RB091            -8.42   RB092             7.25
    1CB102    1RB092             43.   RB093           -2.549
    1CB102    1RB093            -12.
    2CB102    R0000             .095   2RB005            .024
    2CB102    RB006             .055   1RB022             -1.
    2CB102    RB028             .139   RB029             .171
    2CB102    1RB030            .278   RB031             .102
    2CB102    1RB031            .065   RB051             .069
    2CB102    RB061               1.   RB091            -.574
    2CB102    RB092             .493   1RB092             3.9
    2CB102    1RB093            -1.1
    3CB102    RB022              -1.   1RB053              1.
    3CB102    RB102            -8.32   RB103             6.57
    3CB102    RB104              33.   RB105           -2.549
    3CB102    RB106              -3.
    4CB102    1RB022             -1.   1RB053              1.
    4CB102    RB102            -7.66   RB103             5.91
    4CB102    RB104              33.   RB105           -2.549
    4CB102    RB106               3.
    2CB103    R0000             .095   2RB005            .035
    2CB103    RB006              .07   1RB022             -1.
    2CB103    RB028              .15   RB029             .185
    2CB103    1RB030            .231   RB031            -.036
    2CB103    1RB031            .088   RB051             .167
    2CB103    RB061            1.412   RB091           -2.194
    2CB103    RB092            1.999   1RB092             3.3
    2CB103    RB093            -.356   1RB093           -3.09
    2CB104    R0000             .095   2RB005            .035
    2CB104    RB006              .07   1RB022             -1.
    2CB104    RB028              .15   RB029             .185
    2CB104    1RB030            .231   RB031             .085
    2CB104    1RB031            .088   RB051             .046
    2CB104    RB061            1.412   RB091            -.382
    2CB104    RB092             .329   1RB092             2.6
    2CB104    1RB093            -.74
    1CB105    1RB023             -1.   RB046               1.
    1CB105    1RB050              1.   RB087            11.93
    1CB105    1RB087          -14.46   RB088              21.
    1CB105    RB089            -.175   RB090             -25.
    1CB105    1RB090             17.
    1CB106    1RB023             -1.   RB051               1.
    1CB106    RB091              4.4   RB092            -5.57
    1CB106    1RB092            -19.   RB093            -.175
    1CB106    1RB093             13.
    CB111     RB025              -1.   RB053               1.
    CB111     RB094            -1.85   RB095             1.14
    CB111     RB096            -15.8   RB097              66.
    CB112     RB053            -.322   RB054               1.
    CB112     1RB054           -.678
    1CB112    1RB025           -.714   RB053            -.286
    1CB112    RB054               1.
    CB113     RB025              -1.   1RB054              1.
    CB114     2RB025             -1.   1RB054              1.
    1CB114    1RB025             -1.   1RB054              1.
    2CB114    R0000             .197   RB024              -1.
    2CB114    RB034               1.   RB064            4.046
    2CB114    RB065            -5.22   RB066          -11.403
    2CB114    RB067            -.365
    2CB115    R0000             .086   RB024              -1.
    2CB115    RB036               1.   RB068          -11.223
    2CB115    RB069            -5.22   RB070          -11.403
    2CB115    RB071            -.585
    2CB116    RB024              -1.   RB026             .448
    2CB116    RB027             .552   RB031            -.016
    CB117     R0000             .222   RB028              -1.
    CB117     RB034              .99   RB064           -1.174
    CB117     RB065         -71.8799   RB066            56.63
    CB117     RB067            -.411
    2CB117    R0000             .194   1RB024             -1.
    2CB117    RB034               1.   RB064            -.897
    2CB117    RB065            1.282   RB066            -7.81
    2CB117    RB067            -.359
    CB118     R0000             .097   RB028              -1.
    CB118     RB036              .99   RB068          -20.098
    CB118     RB069         -71.8799   RB070            56.63
    CB118     RB071            -.657
    2CB118    R0000             .085   1RB024             -1.
    2CB118    RB036               1.   RB068          -17.179
    2CB118    RB069            1.282   RB070            -7.81
    2CB118    RB071            -.574
    1CB119    R0000              .17   RB029              -1.
    1CB119    RB030              .12   RB034              .88
    1CB119    RB064            1.013   RB065             40.5
    1CB119    RB066          -32.321   RB067            -.315
    2CB119    1RB024             -1.   RB026             .322
    2CB119    RB027             .678   RB031            -.016
    CB120     R0000             .074   RB029              -1.
    CB120     RB030              .12   RB036              .88
    CB120     RB068          -13.612   RB069             40.5
    CB120     RB070          -32.321   RB071            -.504
    2CB120    R0000             .224   RB026              -1.
    2CB120    RB034               1.   RB064             19.1
    2CB120    RB065       -69.606899   RB066           28.206
    2CB120    RB067            -.415
    CB121     R0000              .19   RB029              -1.
    CB121     RB034               1.   RB064             2.39
    CB121     RB065             52.5   RB066            -37.5
    CB121     RB067            -.352
    2CB121    R0000             .098   RB026              -1.
    2CB121    RB036               1.   RB068            6.664
    2CB121    RB069       -69.606899   RB070           28.206
    2CB121    RB071            -.663
    CB122     R0000             .083   RB029              -1.
    CB122     RB036               1.   RB068          -15.125
    CB122     RB069             52.5   RB070            -37.5
    CB122     RB071             -.56
    1CB122    2RB007              1.   RB026              -1.
    1CB122    RB099               1.   UKLDF              -1.
    2CB122    RB026              -1.   RB038               1.
    2CB122    UKLDF              -1.
    CB123     RB030              -1.   RB045               1.
    CB123     RB075            -10.6   RB076             24.3
    CB123     RB077              -8.   RB078             1.75
    CB123     RB079            -1.24   RB080              90.
    1CB123    1RB030             -1.   RB045               1.
    1CB123    RB075             -9.3   RB076             24.6
    1CB123    RB077             10.5   RB078             1.75
    1CB123    RB079            -1.24   RB080              90.
    2CB123    RB026              -1.   RB039               1.
    2CB123    RB072           15.148   RB073          -10.588
    2CB123    RB074             -35.
    CB124     RB030              -1.   RB047               1.
    CB124     RB081             -9.8   RB082             10.6
    CB124     RB083              -8.   RB084             1.75
    CB124     RB085            -1.24   RB086            -3.31
    1CB124    1RB030             -1.   RB047               1.
    1CB124    RB081             -8.5   RB082             10.7
    1CB124    RB083             10.5   RB084             1.75
    1CB124    RB085            -1.24   RB086            -6.87
    2CB124    RB026              -1.   RB031            1.114
    CB125     RB030              -1.   RB046               1.
    CB125     1RB050              1.   RB087           -10.75
    CB125     1RB087            9.25   RB088             -21.
    CB125     RB090             -25.   1RB090             -8.
    1CB125    1RB030             -1.   RB046               1.
    1CB125    1RB050              1.   RB087           -10.26
    1CB125    1RB087            8.76   RB088             -11.
    1CB125    RB090             -25.   1RB090             -8.
    CB126     RB030              -1.   RB051               1.
    CB126     RB091           -18.28   RB092            17.11
    CB126     1RB092            -44.   1RB093            -30.
    1CB126    1RB030             -1.   RB051               1.
    1CB126    RB091           -17.79   RB092            16.62
    1CB126    1RB092            -44.   1RB093            -20.
    2CB126    RB030              -1.   1RB053              1.
    2CB126    RB102           -17.52   RB103            15.77
    2CB126    RB104             -54.   RB106             -15.
    3CB126    1RB030             -1.   1RB053              1.
    3CB126    RB102           -17.03   RB103            15.28
    3CB126    RB104             -54.   RB106              -5.
    CB127     R0000             .201   RB034               1.
    CB127     RB055               1.   RB064            1.066
    CB127     RB065            -4.17   RB066             4.86
    CB127     RB067            -.373
    CB128     R0000             .085   RB036               1.
    CB128     RB055               1.   RB068          -16.666
    CB128     RB069            -4.17   RB070             4.86
    CB128     RB071            -.574
    CB130     R0000             .794   RB064           -11.16
    CB130     RB067               1.
    CB131     R0000             .591   RB068          -11.161
    CB131     RB071               1.
    CB133     R0000             .356   RB027              -.5
    CB133     RB028            -.505   RB033               1.
    1CB133    R0000             .354   RB024             -.23
    1CB133    RB027             -.37   RB028            -.404
    1CB133    RB033               1.
    2CB133    R0000             .355   RB007            -.071
    2CB133    RB027             -.51   RB028            -.424
    2CB133    RB033               1.
    CB135     RB034              -1.   RB035               1.
    CB135     RB064            1.745
    CB138     RB036              -1.   RB037               1.
    CB138     RB068             6.85
    CB139     RB016              -1.   RB031            1.051
    1CB139    1RB016             -1.   RB031            1.049
    CB140     RB031               1.   RB051              -1.
    CB140     RB091              2.1   RB092             -.93
    CB140     1RB093             10.
    CB141     RB038              -1.   1RB038              1.
    CB142     RB047              -1.   2RB047              1.
    CB142     RB083              -2.   RB084             -1.3
    CB142     RB086            -7.42
    1CB142    RB047              -1.   3RB047              1.
    1CB142    RB081               .7   RB082            -27.2
    1CB142    RB083              -2.   RB084             -1.3
    1CB142    RB086            -7.42
    CB143     RB009             -.68   RB010             1.31
    CB143     RB011            -2.32   RB012             1.69
    CB143     RB013             -.46   RB016             -.54
    CB143     1RB044              1.
    1CB143    RB009             -.72   RB010             1.55
    1CB143    RB011            -2.47   RB012             1.65
    1CB143    1RB013            -.38   1RB016            -.63
    1CB143    1RB044              1.
    CB144     4RB045              1.   RB047              -1.
    CB144     RB081              -.8   RB082             -5.4
    CB144     RB083              2.5   RB084              -.5
    CB144     RB086            -7.42
    CB146     1RB001              1.   RB031            1.088
    CB146     RB101             .969
    CB147     RB002               1.   RB031            1.081
    CB147     RB101               1.
    CB149     RB010              -1.   RB031             1.09
    CB150     RB012              -1.   RB031             1.08
    CB151     RB031             1.06   RB047              -1.
    CBF28     RB050             .933   1RB050             -1.
    CBF28     RB051             .067   RB088               2.
    CBF28     1RB090             -5.   RB091            -.514
    CBF28     RB092             .436   1RB092            -2.5
    CBF28     1RB093            -.67
    1CBF29    RB051            -.916   1RB051              1.
    1CBF29    1RB053           -.084   RB091             -.35
    1CBF29    RB092              .38   1RB092           -2.75
    1CBF29    1RB093             5.5   RB102              .28
    1CBF29    RB103             -.26   RB104              .59
    1CBF29    RB106             -.76
    CBF35     RB051            -.344   RB052               1.
    CBF35     1RB053           -.656   RB091               .4
    CBF35     RB092              .14   1RB093            2.58
    CBF35     RB102             2.22   RB103            -2.04
    CBF35     RB104             6.56   RB106            -4.92
    CBF80     RB051              -1.   2RB051              1.
    CBF80     RB092              .41   1RB093              6.
    ADDEA     R0000            .6374   KW1RC              -1.
    CE001     RE001             1.01   RE008            .0018
    CE001     RE010            .0044   RE011            .0122
    CE001     RE012            .0426   RE013             .089
    CE001     RE015             .058   RE017             .093
    CE001     RE018             .117   RE021             .037
    CE001     RE024             .545   RE060           -.0282
    CE001     RE061               1.   1RE061             -1.
    CE001     2RE061              1.   RE062             .208
    CE002     RE002             1.01   RE008            .0015
    CE002     RE010            .0038   RE011            .0114
    CE002     RE012            .0403   RE014             .115
    CE002     RE016             .068   RE017             .115
    CE002     RE019             .136   RE022             .047
    CE002     RE025             .462   RE060           -.0331
    CE002     RE061               1.   1RE061             -1.
    CE002     2RE061              1.   RE062              .24
    CE004     RE004             1.01   RE008            .0013
    CE004     RE010            .0028   RE011            .0125
    CE004     RE012            .0404   RE014             .105
    CE004     RE016             .064   RE017             .101
    CE004     RE019             .082   RE022             .081
    CE004     RE026              .51   RE060           -.0308
    CE004     RE061               1.   1RE061             -1.
    CE004     2RE061              1.   RE062             .208
    CE005     RE008              -.9   RE009               1.
    CE005     RE010              -.1
    CE006     RE008              -1.   RE060             1.19
    CE008     RE008              -1.   RE066               1.
    CE009     RE008               1.   RE010              -1.
    CE010     RE008               1.   RE011              -1.
    CE011     R0000             .108   RE011              -1.
    CE011     RE035               1.   RE068            -2.24
    CE011     RE069         164.1001   RE070            -100.
    CE011     RE071            -.367
    CE012     R0000             .049   RE011              -1.
    CE012     RE037               1.   RE072            -13.1
    CE012     RE073         164.1001   RE074            -100.
    CE012     RE075            -.734
    CE013     R0000             .095   RE012              -1.
    CE013     RE035               1.   RE063               1.
    CE013     RE068            36.97   RE069              55.
    CE013     RE070       -87.899899   RE071            -.323
    CE014     R0000             .044   RE012              -1.
    CE014     RE037               1.   RE063               1.
    CE014     RE072            31.06   RE073              55.
    CE014     RE074       -87.899899   RE075            -.645
    CE015     RE012              -1.   RE039               1.
    CE017     RE012              -1.   RE060             1.11
    CE018     R0000             .113   RE008             .115
    CE018     RE010             .077   RE011             .115
    CE018     RE013              -1.   RE035             .693
    CE018     RE060             -.12   RE063               1.
    CE018     RE068             -.98   RE069             -4.3
    CE018     RE070              -8.   RE071             -.19
    CE019     R0000             .139   RE008             .128
    CE019     RE010             .086   RE011             .127
    CE019     RE013              -1.   RE035             .659
    CE019     RE060            -.126   RE063               1.
    CE019     RE068             -2.5   RE069             -3.1
    CE019     RE070             -3.8   RE071            -.178
    CE020     R0000             .176   RE008             .137
    CE020     RE010             .091   RE011             .137
    CE020     RE013              -1.   RE035             .635
    CE020     RE060             -.13   RE063              1.1
    CE020     RE068            -4.61   RE069             -2.9
    CE020     RE071            -.169
    CE021     R0000             .066   RE008             .097
    CE021     RE010             .065   RE011             .097
    CE021     RE013              -1.   RE037             .741
    CE021     RE060            -.113   RE063              .91
    CE021     RE072            -3.02   RE073             -8.3
    CE021     RE074            -12.7   RE075            -.415
    1CE021    R0000             .049   RE008             .073
    1CE021    RE010              .04   RE011             .063
    1CE021    RE013              -1.   RE037             .824
    1CE021    RE060             -.11   RE063              .91
    1CE021    RE072             7.47   RE073            -16.1
    1CE021    RE074             -2.2   RE075            -.475
    CE022     R0000             .083   RE008             .115
    CE022     RE010             .077   RE011             .115
    CE022     RE013              -1.   RE037             .693
    CE022     RE060             -.12   RE063               1.
    CE022     RE072            -6.33   RE073             -4.3
    CE022     RE074              -8.   RE075             -.38
    CE023     R0000             .116   RE008             .092
    CE023     RE010             .062   RE011             .094
    CE023     RE014              -1.   RE035             .752
    CE023     RE060             -.12   RE063               1.
    CE023     RE068            -1.06   RE069             -6.8
    CE023     RE070             -4.8   RE071            -.204
    CE024     R0000             .151   RE008             .104
    CE024     RE010              .07   RE011             .104
    CE024     RE014              -1.   RE035             .722
    CE024     RE060            -.126   RE063               1.
    CE024     RE068            -2.74   RE069             -5.1
    CE024     RE070             -1.4   RE071            -.194
    CE025     R0000             .177   RE008             .114
    CE025     RE010             .076   RE011             .113
    CE025     RE014              -1.   RE035             .697
    CE025     RE060             -.13   RE063              1.1
    CE025     RE068            -5.02   RE069             -4.4
    CE025     RE070             -1.7   RE071            -.184
    CE026     R0000             .066   RE008             .072
    CE026     RE010             .048   RE011             .081
    CE026     RE014              -1.   RE037             .799
    CE026     RE060            -.113   RE063              .91
    CE026     RE072            -3.21   RE073             -11.
    CE026     RE074             -8.3   RE075            -.441
    1CE026    R0000             .055   RE008             .056
    1CE026    RE010             .032   RE011              .05
    1CE026    RE014              -1.   RE037             .862
    1CE026    RE060             -.11   RE063              .91
    1CE026    RE072             7.67   RE073            -21.2
    1CE026    RE074             -1.1   RE075            -.488
    CE027     R0000             .084   RE008             .092
    CE027     RE010             .062   RE011             .094
    CE027     RE014              -1.   RE037             .752
    CE027     RE060             -.12   RE063               1.
    CE027     RE072            -6.81   RE073             -6.8
    CE027     RE074             -4.8   RE075            -.409
    CE029     RE013              -1.   RE039               1.
    CE032     RE013               1.   RE014              -1.
    CE033     R0000             .136   RE008             .096
    CE033     RE010             .064   RE011             .095
    CE033     RE013             -.62   RE015             -.38
    CE033     RE035             .745   RE060             -.12
    CE033     RE063               1.   RE068             -.67
    CE033     RE069             -8.1   RE070             -1.4
    CE033     RE071            -.204
    CE034     R0000              .18   RE008              .11
    CE034     RE010             .074   RE011             .109
    CE034     RE013             -.62   RE015             -.38
    CE034     RE035             .707   RE060            -.126
    CE034     RE063               1.   RE068            -2.68
    CE034     RE069              -6.   RE070              -.9
    CE034     RE071             -.19
    CE035     R0000             .224   RE008             .124
    CE035     RE010             .074   RE011             .119
    CE035     RE013             -.62   RE015             -.38
    CE035     RE035             .683   RE060             -.13
    CE035     RE063              1.1   RE068            -4.52
    CE035     RE069             -4.8   RE070              4.3
    CE035     RE071            -.182
    CE036     R0000             .079   RE008             .076
    CE036     RE010              .05   RE011             .077
    CE036     RE013             -.62   RE015             -.38
    CE036     RE037             .797   RE060            -.113
    CE036     RE063              .91   RE072            -2.81
    CE036     RE073            -11.4   RE075            -.444
    1CE036    R0000             .062   RE008              .05
    1CE036    RE010             .027   RE011             .043
    1CE036    RE013             -.62   RE015             -.38
    1CE036    RE037              .88   RE060             -.11
    1CE036    RE063              .91   RE072             8.02
    1CE036    RE073            -15.4   RE074             16.8
    1CE036    RE075             -.51
    CE037     R0000             .104   RE008             .096
    CE037     RE010             .064   RE011             .095
    CE037     RE013             -.62   RE015             -.38
    CE037     RE037             .745   RE060             -.12
    CE037     RE063               1.   RE072             -6.6
    CE037     RE073             -8.1   RE074             -1.4
    CE037     RE075            -.407
    CE038     R0000             .137   RE008             .076
    CE038     RE010              .05   RE011             .076
    CE038     RE014            -.625   RE016            -.375
    CE038     RE035             .798   RE060             -.12
    CE038     RE063               1.   RE068             -.71
    CE038     RE069             -13.   RE070               2.
    CE038     RE071            -.215
    CE039     R0000             .181   RE008             .085
    CE039     RE010             .057   RE011             .086
    CE039     RE014            -.625   RE016            -.375
    CE039     RE035             .772   RE060            -.126
    CE039     RE063               1.   RE068             -2.9
    CE039     RE069            -10.2   RE070              3.9
    CE039     RE071            -.206
    CE040     R0000             .183   RE008             .096
    CE040     RE010             .058   RE011             .091
    CE040     RE014            -.625   RE016            -.375
    CE040     RE035             .755   RE060             -.13
    CE040     RE063              1.1   RE068            -4.95
    CE040     RE069             -8.5   RE070              5.6
    CE040     RE071            -.198
    CE041     R0000             .079   RE008             .061
    CE041     RE010             .041   RE011             .062
    CE041     RE014            -.625   RE016            -.375
    CE041     RE037             .836   RE060            -.113
    CE041     RE063              .91   RE072             -2.9
    CE041     RE073            -15.6   RE074              4.3
    CE041     RE075            -.458
    1CE041    R0000             .061   RE008             .039
    1CE041    RE010             .021   RE011             .034
    1CE041    RE014            -.625   RE016            -.375
    1CE041    RE037             .906   RE060             -.11
    1CE041    RE063              .91   RE072             7.99
    1CE041    RE073            -19.1   RE074              9.5
    1CE041    RE075            -.508
    CE042     R0000             .104   RE008             .076
    CE042     RE010              .05   RE011             .076
    CE042     RE014            -.625   RE016            -.375
    CE042     RE037             .798   RE060             -.12
    CE042     RE063               1.   RE072            -6.97
    CE042     RE073             -13.   RE074               2.
    CE042     RE075             -.43
    1CE042    RE015               1.   RE016              -1.
    CE046     RE017              -1.   RE045               1.
    CE046     RE084             3.74   RE085              -9.
    CE046     RE086           -59.07   RE087              .56
    CE047     RE017              -1.   RE048               1.
    CE047     RE088            -29.9   RE089             26.8
    CE047     RE090             -37.   RE091             -93.
    CE047     RE092              .56
    CE049     RE017              -1.   RE058               1.
    CE049     RE098           -23.96   RE099             21.2
    CE049     RE100             -15.   RE101             -83.
    CE049     RE102              .56
    CE051     RE018              -1.   RE045               1.
    CE051     RE084            -1.86   RE085              1.3
    CE051     RE086           -29.85   RE087              -4.
    CE052     RE018              -1.   RE048               1.
    CE052     RE088           -22.22   RE089            19.13
    CE052     RE090            -27.2   RE091             -64.
    CE052     RE092              -4.
    CE054     RE018              -1.   RE058               1.
    CE054     RE098           -16.28   RE099            13.52
    CE054     RE100             -5.2   RE101             -54.
    CE054     RE102              -4.
    CE056     RE019              -1.   RE045               1.
    CE056     RE084            -2.56   RE085             -2.9
    CE056     RE086           -31.85   RE087              -4.
    CE057     RE019              -1.   RE048               1.
    CE057     RE088           -22.22   RE089            19.13
    CE057     RE090            -31.4   RE091             -64.
    CE057     RE092              -4.
    CE059     RE019              -1.   RE058               1.
    CE059     RE098           -16.28   RE099            13.52
    CE059     RE100             -9.4   RE101             -54.
    CE059     RE102              -4.
    CE067     RE021              -1.   RE045               1.
    CE067     RE084            -5.14   RE085              7.3
    CE067     RE086            55.07   RE087            -3.47
    CE068     RE021              -1.   RE048               1.
    CE068     RE088           -16.96   RE089            13.87
    CE068     RE090            -21.2   RE091             -42.
    CE068     RE092            -3.47
    CE070     RE021              -1.   RE058               1.
    CE070     RE098           -11.02   RE099             8.26
    CE070     RE100               .8   RE101             -32.
    CE070     RE102            -3.47
    CE072     RE022              -1.   RE045               1.
    CE072     RE084            -6.58   RE085              3.5
    CE072     RE086        68.049998   RE087            -3.47
    CE073     RE022              -1.   RE048               1.
    CE073     RE088           -16.96   RE089            13.87
    CE073     RE090             -25.   RE091             -31.
    CE073     RE092            -3.47
    CE075     RE022              -1.   RE058               1.
    CE075     RE098           -11.02   RE099             8.26
    CE075     RE100              -3.   RE101             -21.
    CE075     RE102            -3.47
    CE081     RE024              -1.   RE028             .468
    CE081     RE030             .532   RE060            -.027
    CE081     RE064               1.
    CE082     RE024              -1.   RE048               1.
    CE082     RE088               .9   RE089            -3.99
    CE082     RE090               .7   RE091             -17.
    CE082     RE092              -.8
    CE084     RE024              -1.   RE058               1.
    CE084     RE098             6.84   RE099             -9.6
    CE084     RE100             22.7   RE101              -7.
    CE084     RE102              -.8
    CE086     RE025              -1.   RE048               1.
    CE086     RE088             -.73   RE089            -2.36
    CE086     RE090             -16.   RE091               3.
    CE086     RE092              -.8
    CE088     RE025              -1.   RE058               1.
    CE088     RE098             5.21   RE099            -7.97
    CE088     RE100               6.   RE101              13.
    CE088     RE102              -.8
    CE090     RE026              -1.   RE048               1.
    CE090     RE088             1.69   RE089            -4.78
    CE090     RE090             -14.   RE091              -4.
    CE090     RE092              -.8
    CE092     RE026              -1.   RE058               1.
    CE092     RE098             7.63   RE099           -10.39
    CE092     RE100               8.   RE101               6.
    CE092     RE102              -.8
    CE102     RE017            -.147   RE026            -.853
    CE102     1RE056              1.
    CE103     RE019            -.234   RE026            -.766
    CE103     1RE056              1.
    CE104     RE017            -.178   RE026            -.822
    CE104     1RE056              1.
    CE105     RE019            -.284   RE026            -.716
    CE105     1RE056              1.
    CE107     RE028              -1.   RE048               1.
    CE107     RE088            -7.65   RE089             4.56
    CE107     RE090             -15.   RE091              46.
    CE107     RE092              -.8
    CE108     RE048              -1.   RE060               1.
    CE112     RE030              -1.   RE103              .98
    CE124     R0000            .2793   RE068           -9.821
    CE124     RE071               1.
    CE125     R0000            .2564   RE072           -15.27
    CE125     RE075               1.
    CE126     RE035              -1.   RE036               1.
    CE126     RE068            -1.31   RE069            -5.23
    CE130     RE037              -1.   RE040               1.
    CE130     RE072            -4.73   RE073             -5.4
    CE134     RE045              -1.   RE046               1.
    CE135     RE045              -1.   RE047               1.
    CE135     RE085               7.
    CE136     RE048              -1.   RE049               1.
    CE136     RE088             5.94   RE089             -5.6
    CE136     RE090               7.   RE091              21.
    CE137     RE048              -1.   RE050               1.
    CE137     RE088             4.12   RE089            -2.85
    CE137     RE090               7.   RE091              21.
    CE138     RE017             -.06   RE022             -.13
    CE138     RE025             -.81   RE051               1.
    CE142     RE055               1.   RE058              -1.
    CE142     RE098            -1.82   RE099             2.76
    CE142     RE100              -5.   RE101               8.
    CE143     RE017              .13   RE044              -.4
    CE143     RE057               1.   RE058             -.73
    CH002     RH001               1.   1RH001            1.01
    CH002     RH007            .0024   RH008            .0099
    CH002     1RH008           .0022   RH010            .0382
    CH002     RH016            .1081   RH018            .0242
    CH002     RH019             .101   RH020            .0929
    CH002     2RH025            .005   RH026            .6093
    CH002     RH096            -.017   RH107               1.
    CH002     1RH107             -1.   RH113            .0048
    1CH002    R0000             .001   RH001               3.
    1CH002    RH106              -1.
    2CH002    RH001            1.111   1RH001            1.01
    2CH002    RH007            .0024   RH008            .0099
    2CH002    1RH008           .0022   RH010            .0432
    2CH002    RH016            .1011   RH018            .0242
    2CH002    RH019             .103   RH020            .1029
    2CH002    2RH025            .005   RH026            .6063
    2CH002    RH028            -.007   RH096            -.016
    2CH002    RH107               1.   1RH107             -1.
    2CH002    RH113            .0048
    3CH002    RH001            1.126   1RH001            .505
    3CH002    RH003             .505   RH007            .0031
    3CH002    RH008            .0095   1RH008           .0013
    3CH002    RH010             .049   RH016            .1182
    3CH002    RH018            .0283   RH019            .1182
    3CH002    RH020            .1101   2RH025            .005
    3CH002    RH026             .304   RH027            .2556
    3CH002    RH028            -.007   RH096            -.016
    3CH002    RH107            1.126   1RH107             -1.
    3CH002    RH113            .0027
    4CH002    RH001               1.   1RH001            1.01
    4CH002    RH007            .0024   RH008            .0099
    4CH002    1RH008           .0022   RH010            .0382
    4CH002    RH016            .1081   RH018            .0242
    4CH002    RH019             .101   RH020            .0929
    4CH002    2RH025            .005   RH026            .6093
    4CH002    RH096            -.017   RH113            .0048
    CH003     RH001              1.1   RH002             1.01
    CH003     RH007            .0024   RH008             .009
    CH003     1RH008           .0019   RH010            .0259
    CH003     RH015            .1192   RH017            .5324
    CH003     RH018            .0354   2RH018           .1273
    CH003     RH021            .1333   2RH025            .005
    CH003     RH096           -.0145   RH107              1.1
    CH003     1RH107             -1.   RH113            .0062
    1CH003    1RH001              1.   RH096               1.
    2CH003    RH003               1.   RH096               1.
    CH004     RH001            1.027   RH003             1.01
    CH004     RH007            .0039   RH008            .0092
    CH004     1RH008           .0004   RH010            .0485
    CH004     RH016            .1344   RH018            .0394
    CH004     RH019            .1273   RH020             .107
    CH004     2RH025            .005   RH027            .5223
    CH004     RH096            -.015   RH113            .0006
    1CH004    RH001            1.027   RH003             1.01
    1CH004    RH007            .0039   RH008            .0092
    1CH004    1RH008           .0004   RH010            .0485
    1CH004    RH016            .1344   RH018            .0394
    1CH004    RH019            .1373   RH020             .107
    1CH004    2RH025            .005   RH027            .5223
    1CH004    RH096            -.015   RH107               1.
    1CH004    1RH107             -1.   RH113            .0006
    CH005     RH001            1.067   RH004             1.01
    CH005     RH007            .0015   RH008            .0049
    CH005     1RH008           .0009   RH010            .0137
    CH005     RH015            .0687   1RH017           .6486
    CH005     RH018            .0253   1RH019            .101
    CH005     RH024            .1243   2RH025            .005
    CH005     RH096            -.017   RH107            1.067
    CH005     RH113            .0041
    1CH005    RH001            1.067   RH004             1.01
    1CH005    RH007            .0015   RH008            .0049
    1CH005    1RH008           .0009   RH010            .0137
    1CH005    RH015            .0687   1RH017           .3684
    1CH005    RH018            .0253   1RH019            .101
    1CH005    RH024            .1243   2RH025            .005
    1CH005    1RH028           .0168   RH029            .1107
    1CH005    1RH030           .1527   1RH060           .2802
    1CH005    RH096            -.024   RH107            1.067
    1CH005    RH113            .0041
    2CH005    RH001              1.1   1RH005            1.01
    2CH005    RH007            .0002   RH008            .0004
    2CH005    1RH008           .0003   RH010            .0236
    2CH005    RH015              .07   RH018              .03
    2CH005    RH019              .15   RH023              .22
    2CH005    2RH025            .005   2RH026              .5
    2CH005    RH096            -.017   RH107              1.1
    2CH005    1RH107             -1.   RH113            .0005
    4CH005    RH001              1.1   RH006             1.01
    4CH005    RH007             .002   RH008            .0072
    4CH005    1RH008           .0005   RH010            .0378
    4CH005    RH016            .1495   2RH017           .5009
    4CH005    RH018            .0436   3RH018           .1228
    4CH005    RH021            .1257   2RH025            .005
    4CH005    RH096            -.017   RH107              1.1
    4CH005    1RH107             -1.   RH113             .003
    CH006     RH001              1.1   RH006             1.01
    CH006     RH007             .002   RH008            .0072
    CH006     1RH008           .0005   RH010            .0378
    CH006     RH016            .1495   2RH017           .5009
    CH006     RH018            .0436   3RH018           .1228
    CH006     RH021            .1257   2RH025            .005
    CH006     RH096            -.017   RH113             .003
    1CH006    R0000             .001   1RH008              1.
    1CH006    RH055               1.   RH114              -1.
    2CH006    RH001           15.401   RH006            14.14
    2CH006    RH007             .028   RH008             .101
    2CH006    1RH008            .007   RH010             .529
    2CH006    RH016            2.093   RH018              .61
    2CH006    3RH018           1.719   RH021             1.76
    2CH006    4RH022           1.777   2RH025             .07
    2CH006    RH028             .933   RH029            2.195
    2CH006    3RH030           1.088   RH050               1.
    2CH006    RH060            7.062   RH096            -.641
    2CH006    RH113             .042
    CH007     RH007              -1.   RH008               1.
    CH008     R0000              .38   RH008              -1.
    CH008     RH034               1.   RH063            4.665
    CH008     RH064          -89.655   RH065       238.949997
    CH008     RH066      -206.899897   1RH066           -.579
    CH008     RH096            -.002
    CH009     R0000             .247   RH008              -1.
    CH009     RH035               1.   RH067           -1.379
    CH009     RH068          -89.655   RH069       238.949997
    CH009     RH070      -206.899897   RH071            -.425
    CH009     RH096            -.002
    CH010     R0000             .114   RH008              -1.
    CH010     RH036               1.   RH072          -22.586
    CH010     RH073       187.751997   RH074              95.
    CH010     RH075            -.772   RH096            -.002
    CH012     RH008              -1.   RH114               1.
    CH013     R0000             .026   RH007            -.544
    CH013     RH009             -.49   RH010             -.07
    CH013     RH014               1.   RH059               1.
    CH013     RH096            -.146   RH114             .033
    CH014     RH008               1.   RH009              -1.
    CH015     R0000             .353   RH010              -1.
    CH015     RH034               1.   RH054               1.
    CH015     RH063              .96   RH064            -99.2
    CH015     RH065       132.080097   RH066            -128.
    CH015     1RH066           -.538   RH096            -.222
    CH016     R0000             .229   RH010              -1.
    CH016     RH035               1.   RH054               1.
    CH016     RH067             -3.2   RH068            -99.2
    CH016     RH069       132.080097   RH070            -128.
    CH016     RH071            -.394   RH096            -.222
    CH017     R0000             .102   RH010              -1.
    CH017     RH036               1.   RH072           18.154
    CH017     RH073           58.074   RH074              -5.
    CH017     RH075            -.689   RH096            -.003
    CH018     RH010              -1.   RH039               1.
    CH018     UKLDF              -1.
    1CH019    R0000             .221   RH010              -1.
    1CH019    RH035               1.   RH067            29.23
    1CH019    RH068             -80.   RH069            61.13
    1CH019    RH070           -30.77   RH071             -.38
    1CH019    RH096            -.003
    CH020     RH011               1.   2RH038             -1.
    CH021     RH011               1.   RH038              -1.
    CH022     RH023              -1.   RH048               1.
    CH022     RH085           -7.833   RH086           -14.97
    CH022     RH087             -8.4   RH088              13.
    CH022     RH089           -2.729
    1CH022    RH023              -1.   RH043               1.
    1CH022    RH080           -10.08   RH081             4.88
    1CH022    RH082             -8.4   RH083              13.
    1CH022    RH084           -2.729
    2CH022    RH023              -1.   RH046               1.
    2CH022    RH062           -.0014   1RH089           -5.67
    2CH022    2RH089             14.   3RH089          -20.07
    CH024     R0000             .092   RH036               1.
    CH024     RH072           -5.587   RH073            4.268
    CH024     RH074              -5.   RH075            -.626
    CH024     RH095              -1.
    1CH024    R0000               .2   RH035               1.
    1CH024    RH067           10.475   RH068          -33.519
    1CH024    RH069             7.06   RH070            -6.98
    1CH024    RH071            -.344   RH095              -1.
    CH025     RH095              -1.   RH096             1.11
    1CH025    R0000             .308   RH034               1.
    1CH025    RH063           13.966   RH064          -33.519
    1CH025    RH065             7.06   RH066            -6.98
    1CH025    1RH066           -.469   RH095              -1.
    CH026     R0000             .254   RH011              -1.
    CH026     RH034               1.   RH063          -10.957
    CH026     RH064          78.4309   RH065           -46.88
    CH026     RH066             34.6   1RH066           -.387
    CH027     R0000             .165   RH011              -1.
    CH027     RH035               1.   RH067           -15.34
    CH027     RH068          78.4309   RH069           -46.88
    CH027     RH070             34.6   RH071            -.284
    CH028     R0000             .324   RH012              -1.
    CH028     RH034               1.   RH063            5.294
    CH028     RH064         -69.1179   RH065           44.198
    CH028     RH066           -26.47   1RH066           -.494
    CH028     RH096            -.003
    CH029     R0000             .211   RH012              -1.
    CH029     RH035               1.   RH067            -1.03
    CH029     RH068         -69.1179   RH069           44.198
    CH029     RH070           -26.47   RH071            -.362
    CH029     RH096            -.003
    CH030     R0000             .097   RH012              -1.
    CH030     RH036               1.   RH072          -20.147
    CH030     RH073           41.244   RH074              -5.
    CH030     RH075            -.659   RH096            -.003
    CH031     R0000              .18   RH013              -1.
    CH031     RH035               1.   RH067            3.015
    CH031     RH068           45.226   RH069           -50.89
    CH031     RH070            25.13   RH071             -.31
    CH031     RH096            -.002
    CH032     R0000             .083   RH013              -1.
    CH032     RH036               1.   RH072          -15.201
    CH032     RH073           -53.41   RH074              -5.
    CH032     RH075            -.563   RH096            -.002
    CH033     R0000            1.039   RH010            -.124
    CH033     RH014            -.709   RH031               1.
    CH033     RH054             .124   RH095            -.167
    CH033     RH096            -.031
    CH034     R0000             .662   RH010            -.122
    CH034     RH014            -.472   RH032               1.
    CH034     RH054             .122   RH095            -.406
    CH034     RH096            -.038
    CH035     R0000             .315   RH014              -1.
    CH035     RH034               1.   RH064            1.429
    CH035     RH065           -24.49   RH066              -5.
    CH035     1RH066            -.48
    CH036     R0000             .205   RH014              -1.
    CH036     RH035               1.   RH067           -3.143
    CH036     RH068            1.429   RH069           -24.49
    CH036     RH070              -5.   RH071            -.352
    1CH036    R0000             .591   RH063            -6.25
    1CH036    1RH066              1.
    CH037     R0000             .591   RH067          -10.268
    CH037     RH071               1.
    CH038     R0000             .591   RH072           -11.16
    CH038     RH075               1.
    CH039     R0000              .09   RH016              -1.
    CH039     RH036               1.   RH072           27.459
    CH039     RH073          -36.543   RH074              -5.
    CH039     RH075            -.612   RH096            -.002
    CH040     RH010             -.25   RH016             -.75
    CH040     RH039               1.   UKLDF              -1.
    CH042     RH010              -1.   2RH039              1.
    CH042     1RH106              1.   UKLDF              -1.
    CH043     RH010              -.7   RH016              -.3
    CH043     2RH039              1.   1RH106              1.
    CH043     UKLDF              -1.
    CH044     R0000             .089   RH015              -1.
    CH044     RH036               1.   RH072           25.706
    CH044     RH073          -36.248   RH074              -5.
    CH044     RH075            -.603   RH096            -.002
    CH045     RH010             -.25   RH015             -.75
    CH045     RH039               1.   UKLDF              -1.
    CH046     RH010              -.7   RH015              -.3
    CH046     2RH039              1.   1RH106              1.
    CH046     UKLDF              -1.
    CH048     R0000              .12   RH007             .029
    CH048     RH008             .043   RH016              -1.
    CH048     RH036             .755   RH057             .883
    CH048     RH072           -8.657   RH073          -12.314
    CH048     RH074           -2.265   RH075            -.451
    CH048     RH096             -.12   RH113             .087
    CH048     RH114             .086
    CH050     R0000             .364   RH007             .042
    CH050     RH008             .071   RH016              -1.
    CH050     RH034             .621   RH057            1.533
    CH050     RH058               1.   RH063           -1.611
    CH050     RH064            4.832   RH065           -4.213
    CH050     RH066             2.42   1RH066           -.271
    CH050     RH096            -.133   RH113             .133
    CH050     RH114             .133
    CH051     R0000             .304   RH007             .042
    CH051     RH008             .071   RH016              -1.
    CH051     RH035             .621   RH057            1.533
    CH051     RH058               1.   RH067           -4.026
    CH051     RH068            4.832   RH069           -4.213
    CH051     RH070             2.42   RH071            -.198
    CH051     RH096            -.133   RH113             .133
    CH051     RH114             .133
    CH056     R0000             .123   RH007             .023
    CH056     RH008             .031   RH016             -.75
    CH056     RH018             -.25   RH036             .806
    CH056     RH057             .872   RH072           -8.685
    CH056     RH073          -15.602   RH074           -2.418
    CH056     RH075            -.474   RH096             -.12
    CH056     RH113              .07   RH114              .07
    CH057     R0000             .306   RH007              .03
    CH057     RH008             .044   RH016             -.75
    CH057     RH018             -.25   RH034              .73
    CH057     RH057             .982   RH063            3.189
    CH057     RH064            4.832   RH065           -10.03
    CH057     RH066              5.8   1RH066           -.325
    CH057     RH096             -.12   RH113             .098
    CH057     RH114             .098
    CH058     R0000             .233   RH007              .03
    CH058     RH008             .044   RH016             -.75
    CH058     RH018             -.25   RH035              .73
    CH058     RH057             .982   RH067             .097
    CH058     RH068            4.832   RH069           -10.03
    CH058     RH070              5.8   RH071            -.238
    CH058     RH096             -.12   RH113             .098
    CH058     RH114             .098
    CH062     R0000             .295   RH007             .023
    CH062     RH008              .03   RH015             -.74
    CH062     RH018             -.26   RH034             .776
    CH062     RH057             .963   RH063            3.371
    CH062     RH064            8.683   RH065           -15.57
    CH062     RH066             7.15   1RH066           -.343
    CH062     RH096            -.114   RH113             .086
    CH062     RH114             .085
    CH063     R0000             .219   RH007             .023
    CH063     RH008              .03   RH015             -.74
    CH063     RH018             -.26   RH035             .776
    CH063     RH057             .963   RH067             .102
    CH063     RH068            8.683   RH069           -15.57
    CH063     RH070             7.15   RH071            -.252
    CH063     RH096            -.114   RH113             .086
    CH063     RH114             .085
    CH065     R0000              .18   RH007              .04
    CH065     RH008              .06   RH010              -.3
    CH065     RH016              -.7   1RH038             .66
    CH065     RH057             1.15   RH096             -.12
    CH065     RH113              .12   RH114              .12
    CH066     R0000             .099   RH007             .035
    CH066     RH008             .055   RH016              -1.
    CH066     1RH038            .684   RH057               1.
    CH066     RH096             -.12   RH113             .113
    CH066     RH114             .113
    CH067     R0000             .349   RH007             .042
    CH067     RH008             .071   RH011             .214
    CH067     RH016              -1.   RH034              .36
    CH067     RH038             .019   2RH038            .028
    CH067     RH057            1.533   RH058               1.
    CH067     RH063            2.675   RH064          -11.675
    CH067     RH065            2.457   RH066            -2.43
    CH067     1RH066           -.163   RH096            -.133
    CH067     RH113             .133   RH114             .133
    CH068     R0000             .262   RH007             .042
    CH068     RH008             .071   RH011             .214
    CH068     RH016              -1.   RH035              .36
    CH068     RH038             .019   2RH038            .028
    CH068     RH057            1.533   RH058               1.
    CH068     RH067             .973   RH068          -11.676
    CH068     RH069            2.457   RH070            -2.43
    CH068     RH071            -.119   RH096            -.133
    CH068     RH113             .133   RH114             .133
    CH069     R0000             .079   RH007             .028
    CH069     RH008              .04   RH015              -1.
    CH069     1RH038            .735   RH057             .977
    CH069     RH096            -.114   RH113             .098
    CH069     RH114             .099
    CH070     R0000             .294   RH007             .034
    CH070     RH008             .054   RH011             .236
    CH070     RH015              -1.   RH034             .395
    CH070     RH038             .019   2RH038             .03
    CH070     RH057            1.498   RH058             .977
    CH070     RH063            2.936   RH064          -12.811
    CH070     RH065              2.7   RH066            -2.67
    CH070     1RH066           -.179   RH096            -.126
    CH070     RH113             .116   RH114             .116
    CH071     R0000             .264   RH007             .034
    CH071     RH008             .054   RH011             .236
    CH071     RH015              -1.   RH035             .395
    CH071     RH038             .019   2RH038             .03
    CH071     RH057            1.498   RH058             .977
    CH071     RH067            1.067   RH068           -12.81
    CH071     RH069              2.7   RH070            -2.67
    CH071     RH071            -.131   RH096            -.126
    CH071     RH113             .116   RH114             .116
    CH072     R0000             .098   RH007              .03
    CH072     RH008             .044   RH011              .26
    CH072     RH016             -.75   RH018             -.25
    CH072     RH038             .014   2RH038            .023
    CH072     RH057             .982   RH095             .433
    CH072     RH096             -.12   RH113             .098
    CH072     RH114             .098
    CH074     R0000              .15   RH007             .035
    CH074     RH008             .055   RH011             .142
    CH074     RH016              -1.   RH038             .051
    CH074     RH057            1.359   RH095             .491
    CH074     RH096            -.171   RH113             .113
    CH074     RH114             .113
    1CH074    R0000             .133   RH007             .035
    1CH074    RH008             .055   RH011             .104
    1CH074    RH016              -1.   RH038             .038
    1CH074    2RH038             .06   RH057             1.21
    1CH074    RH095             .482   RH096             -.15
    1CH074    RH113             .113   RH114             .113
    2CH074    R0000             .133   RH007             .035
    2CH074    RH008             .055   RH011              .17
    2CH074    RH016              -1.   RH038             .032
    2CH074    RH057             1.21   RH095             .482
    2CH074    RH096             -.15   RH113             .113
    2CH074    RH114             .113
    CH075     R0000             .121   RH007             .028
    CH075     RH008              .04   RH011              .16
    CH075     RH015              -1.   RH038             .052
    CH075     RH057            1.336   RH095             .523
    CH075     RH096            -.163   RH113             .098
    CH075     RH114             .099
    1CH075    R0000             .108   RH007             .028
    1CH075    RH008              .04   RH011              .12
    1CH075    RH015              -1.   RH038             .039
    1CH075    2RH038            .062   RH057            1.188
    1CH075    RH095             .514   RH096            -.142
    1CH075    RH113             .099   RH114             .098
    2CH075    R0000             .108   RH007             .028
    2CH075    RH008              .04   RH011             .188
    2CH075    RH015              -1.   RH038             .033
    2CH075    RH057            1.188   RH095             .514
    2CH075    RH096            -.142   RH113             .099
    2CH075    RH114             .098
    CH076     R0000             .078   RH007             .023
    CH076     RH008              .03   RH011             .257
    CH076     RH015             -.74   RH018             -.26
    CH076     RH038             .015   2RH038            .024
    CH076     RH057             .963   RH095              .48
    CH076     RH096            -.114   RH113             .086
    CH076     RH114             .085
    CH077     R0000             .284   RH034               1.
    CH077     1RH038             -1.   RH063            3.732
    CH077     RH064             .644   RH065          -10.936
    CH077     RH066             6.43   1RH066           -.432
    1CH077    R0000             .185   RH035               1.
    1CH077    1RH038             -1.   RH067             .257
    1CH077    RH068             .643   RH069          -10.936
    1CH077    RH070             6.43   RH071            -.317
    CH078     RH011             .268   RH038             .024
    CH078     1RH038             -1.   2RH038            .038
    CH078     RH095              .67
    1CH078    RH011              .31   RH038              .02
    1CH078    1RH038             -1.   RH095              .67
    3CH078    1RH025           -1.02   RH046               1.
    3CH078    RH062             .007   1RH089          -3.481
    3CH078    2RH089            47.3   3RH089         -31.567
    3CH078    RH096            -.023   RH113             .013
    4CH078    RH019            -1.02   RH046               1.
    4CH078    RH062            .0002   1RH089           6.108
    4CH078    2RH089            -13.   3RH089         -34.293
    4CH078    RH096            -.023   RH113             .013
    CH079     RH019            -1.02   RH046               1.
    CH079     RH062           -.0008   1RH089           4.409
    CH079     2RH089            -9.7   3RH089         -33.827
    CH079     RH096            -.023   RH113             .013
    1CH079    1RH019           -1.02   RH046               1.
    1CH079    RH062           -.0013   1RH089           4.257
    1CH079    2RH089            -5.5   3RH089         -33.785
    1CH079    RH096            -.023   RH113             .013
    2CH079    RH020            -1.02   RH046               1.
    2CH079    RH062            .0083   1RH089           1.871
    2CH079    2RH089           -11.1   3RH089         -15.719
    2CH079    RH096            -.023   RH113             .013
    3CH079    RH021            -1.02   RH046               1.
    3CH079    RH062            .0055   1RH089            .569
    3CH079    2RH089            -7.1   3RH089          -15.55
    3CH079    RH096            -.023   RH113             .013
    4CH079    RH024            -1.02   RH046               1.
    4CH079    RH062            .0048   1RH089          -1.677
    4CH079    2RH089             -.8   3RH089         -11.972
    4CH079    RH096            -.023   RH113             .013
    CH080     RH028            -1.02   RH046               1.
    CH080     RH062            .0155   1RH089          -1.677
    CH080     2RH089             -.4   3RH089           2.934
    CH080     RH096            -.023   RH113             .013
    1CH080    1RH028           -1.02   RH046               1.
    1CH080    RH062             .013   1RH089          -3.039
    1CH080    2RH089             6.1   3RH089             2.9
    1CH080    RH096            -.023   RH113             .013
    2CH080    RH028              -.8   RH029              -.2
    2CH080    RH046               1.   RH062            .0162
    2CH080    1RH089          -1.951   2RH089             .45
    2CH080    3RH089           28.92   RH096            -.023
    2CH080    RH113             .013
    3CH080    1RH028           -.612   RH029            -.408
    3CH080    RH046               1.   RH062             .014
    3CH080    1RH089          -3.575   2RH089             6.7
    3CH080    3RH089          54.272   RH096            -.023
    3CH080    RH113             .013
    4CH080    RH025            -1.02   RH046               1.
    4CH080    RH062             .022   1RH089          -7.066
    4CH080    2RH089            33.2   3RH089         -11.422
    4CH080    RH096            -.023   RH113             .013
    CH081     RH025            -.714   RH046               1.
    CH081     RH052            -.306   RH062             .023
    CH081     1RH089          -7.813   2RH089            33.8
    CH081     3RH089           9.455   RH093            6.343
    CH081     1RH093             5.5   RH094            3.978
    CH081     RH096            -.023   RH110           -6.138
    CH081     RH113             .013
    1CH081    1RH021           -1.02   RH046               1.
    1CH081    RH062            .0025   1RH089           -.705
    1CH081    2RH089            -1.5   3RH089         -13.728
    1CH081    RH096            -.023   RH113             .013
    CH086     RH018              -.2   RH019              -.8
    CH086     1RH025              .2   RH041               .8
    CH086     RH056               1.   RH096            -.021
    1CH086    1RH025             -1.   RH043               1.
    1CH086    RH080           -9.171   RH081           -6.286
    1CH086    RH083              49.   RH084            5.623
    CH087     1RH025             -1.   RH048               1.
    CH087     RH085           -6.926   RH086          -26.286
    CH087     RH088              49.   RH089            5.623
    CH089     RH018              -.2   RH019              -.8
    CH089     RH043               1.   RH080            2.487
    CH089     RH081           -6.927   RH082             -7.4
    CH089     RH083             -15.   RH084            7.987
    1CH089    RH018             -.18   2RH018            -.82
    1CH089    RH043               1.   RH080            1.076
    1CH089    RH081           -6.849   RH082              -8.
    1CH089    RH083             -14.   RH084            7.987
    2CH089    RH018             -.18   1RH019            -.82
    2CH089    RH043               1.   RH080             .921
    2CH089    RH081           -6.841   RH082             -8.5
    2CH089    RH083              -8.   RH084            7.987
    CH090     RH018              -.2   RH019              -.8
    CH090     RH048               1.   RH085            4.782
    CH090     RH086          -28.967   RH087             -7.4
    CH090     RH088             -15.   RH089            7.987
    1CH090    RH018             -.18   2RH018            -.82
    1CH090    RH048               1.   RH085            3.321
    1CH090    RH086          -28.643   RH087              -8.
    1CH090    RH088             -14.   RH089            7.987
    2CH090    RH018             -.18   1RH019            -.82
    2CH090    RH048               1.   RH085            3.166
    2CH090    RH086          -28.607   RH087             -8.5
    2CH090    RH088              -8.   RH089            7.987
    CH093     RH018             -.22   2RH018            -.78
    CH093     1RH025              .2   RH041               .8
    CH093     RH056            1.488   RH096            -.021
    1CH093    RH018             -.22   3RH018            -.78
    1CH093    1RH025              .2   RH041               .8
    1CH093    RH056            1.488   RH096            -.021
    CH097     RH019              -1.   RH043               1.
    CH097     RH080            1.699   RH081           -6.884
    CH097     RH082             -6.8   RH083             -14.
    CH097     RH084            2.987
    1CH097    2RH018             -1.   RH043               1.
    1CH097    RH080             .152   RH081           -6.799
    1CH097    RH082             -7.8   RH083             -11.
    1CH097    RH084            2.987
    2CH097    1RH019             -1.   RH043               1.
    2CH097    RH081            -6.79   RH082             -8.3
    2CH097    RH083             -6.5   RH084            2.987
    CH098     RH019              -1.   RH048               1.
    CH098     RH085            3.944   RH086          -28.786
    CH098     RH087             -6.8   RH088             -14.
    CH098     RH089            2.987
    1CH098    2RH018             -1.   RH048               1.
    1CH098    RH085            2.397   RH086           -28.43
    1CH098    RH087             -7.8   RH088             -11.
    1CH098    RH089            2.987
    2CH098    1RH019             -1.   RH048               1.
    2CH098    RH085            2.244   RH086          -28.395
    2CH098    RH087             -8.3   RH088             -6.5
    2CH098    RH089            2.987
    CH101     RH020              -1.   RH043               1.
    CH101     RH080           -3.697   RH081           10.778
    CH101     RH082              1.3   RH083              -9.
    CH101     RH084           -2.795
    1CH101    RH020              -1.   RH043              .98
    1CH101    RH062            .0098   RH080           -2.488
    1CH101    RH081           10.665   RH082            -8.33
    1CH101    RH083           -11.76   RH084           -2.674
    1CH101    RH096            -.023   RH113             .013
    CH102     RH020              -1.   RH048               1.
    CH102     RH085           -1.452   RH086           -10.18
    CH102     RH087              1.3   RH088              -9.
    CH102     RH089           -2.795
    1CH102    RH020              -1.   RH048              .98
    1CH102    RH062            .0098   RH085            -.288
    1CH102    RH086          -10.073   RH087            -8.33
    1CH102    RH088           -11.76   RH089           -2.674
    1CH102    RH096            -.023   RH113             .013
    CH105     RH021              -1.   RH043               1.
    CH105     RH080           -4.551   RH081           10.702
    CH105     RH082             -1.5   RH083              -6.
    CH105     RH084           -2.729
    CH106     RH021              -1.   RH048               1.
    CH106     RH085           -2.306   RH086          -10.107
    CH106     RH087             -1.5   RH088             -4.5
    CH106     RH089           -2.729
    CH112     1RH021             -1.   RH043               1.
    CH112     RH080           -5.532   RH081           12.264
    CH112     RH082             -4.5   RH083             -2.5
    CH112     RH084           -2.729
    CH113     1RH021             -1.   RH048               1.
    CH113     RH085           -3.287   RH086           -8.373
    CH113     RH087             -4.5   RH088             -2.5
    CH113     RH089           -2.729
    CH114     RH024              -1.   RH043               1.
    CH114     RH080            -6.77   RH081           13.769
    CH114     RH082             -2.2   RH084           -2.729
    CH115     RH024              -1.   RH048               1.
    CH115     RH085           -4.526   RH086           -6.651
    CH115     RH087             -2.2   RH089           -2.729
    CH119     RH026              -1.   RH028              .07
    CH119     RH029             .459   RH030             .471
    CH119     RH060               1.   RH096            -.027
    2CH119    2RH017             -1.   RH028              .08
    2CH119    RH029             .537   3RH030            .383
    2CH119    RH060               1.   RH096            -.027
    CH121     RH027              -1.   RH028             .113
    CH121     RH029             .525   RH030             .362
    CH121     RH060            1.007   RH096            -.027
    CH122     4RH022           1.777   RH026           -8.174
    CH122     RH028             .572   RH029            2.195
    CH122     RH030             2.61   RH050               1.
    CH122     RH060            8.174   RH096            -.651
    CH124     4RH022           1.777   RH027           -7.147
    CH124     RH028             .808   RH029            2.195
    CH124     RH030            1.347   RH050               1.
    CH124     RH060              7.2   RH096            -.624
    CH126     RH028              -1.   RH048               1.
    CH126     RH085           -6.005   RH086            8.065
    CH126     RH087              8.5   RH088              4.5
    CH126     RH089           -3.281
    1CH126    1RH028             -1.   RH048               1.
    1CH126    RH085           -6.926   RH086               8.
    1CH126    RH087               6.   RH088              10.
    1CH126    RH089           -3.281
    2CH126    1RH028             -.6   RH029              -.4
    2CH126    RH048               1.   RH085           -7.576
    2CH126    RH086           58.523   RH087               7.
    2CH126    RH088              11.   RH089            -2.95
    3CH126    RH028              -.8   RH029              -.2
    3CH126    RH048               1.   RH085           -6.334
    3CH126    RH086              21.   RH087             9.25
    3CH126    RH088             5.25   RH089           -2.953
    CH128     RH028              -1.   RH043               1.
    CH128     RH080            -8.25   RH081           28.226
    CH128     RH082              8.5   RH083              4.5
    CH128     RH084           -3.281
    1CH128    1RH028             -1.   RH043               1.
    1CH128    RH080           -9.171   RH081              28.
    1CH128    RH082               6.   RH083              10.
    1CH128    RH084           -3.281
    1CH130    R0000              .15   RH007             .022
    1CH130    RH008             .007   RH009             .037
    1CH130    RH012             .157   RH013             .119
    1CH130    RH025             .222   2RH025            .322
    1CH130    RH029              -1.   RH061               1.
    1CH130    RH096            -.047   RH114             .069
    1CH131    R0000              .15   RH007             .028
    1CH131    RH008             .009   RH009             .049
    1CH131    RH012             .176   RH013             .122
    1CH131    RH025             .227   2RH025            .268
    1CH131    RH029              -1.   RH061            1.138
    1CH131    RH096            -.047   RH114             .073
    2CH131    R0000              .15   RH007             .035
    2CH131    RH008             .011   RH009              .06
    2CH131    RH012             .195   RH013             .126
    2CH131    RH025             .228   2RH025            .216
    2CH131    RH029              -1.   RH061            1.222
    2CH131    RH096            -.047   RH114             .077
    CH134     RH025              -1.   RH043               1.
    CH134     RH080          -14.168   RH081           12.896
    CH134     RH082              15.   RH083             40.5
    CH134     RH084           -2.729
    CH135     RH025              -1.   RH048               1.
    CH135     RH085          -11.923   RH086            -6.23
    CH135     RH087              15.   RH088             40.5
    CH135     RH089           -2.729
    1CH140    RH048              -1.   2RH048              1.
    1CH140    RH085            -.739   RH086            6.951
    1CH140    RH087               3.   RH088              -5.
    2CH140    RH048              -1.   3RH048              1.
    2CH140    RH085            -.739   RH086           -5.732
    2CH140    RH087               3.   RH088              -5.
    CH143     1RH018             -1.   RH096               1.
    CH144     RH020              -1.   RH096               1.
    CH145     RH025              -1.   RH096               1.
    CH146     RH028              -1.   RH096               1.
    CH150     RH039              -1.   RH096             1.11
    CH150     UKLDF               1.
    CH151     RH052              -1.   RH096               1.
    1CH151    RH029              -1.   RH096               1.
    2CH151    RH013              -1.   RH096              1.1
    CH153     RH042              -1.   2RH042              1.
    CH155     RH018             -.49   3RH018            -.51
    CH155     RH042               1.   RH096            -.002
    CH156     RH018             -.64   RH019             -.36
    CH156     RH042               1.   RH096            -.002
    1CH156    1RH018              1.   2RH018             -1.
    2CH156    2RH018              1.   3RH018             -1.
    3CH156    1RH018              1.   RH019              -1.
    4CH156    1RH018              1.   1RH019             -1.
    CH180     RH001            1.151   2RH001            1.01
    CH180     RH007            .0022   RH008            .0088
    CH180     1RH008            .002   RH010             .065
    CH180     RH016             .157   RH018              .03
    CH180     RH019              .09   RH020             .083
    CH180     2RH025            .005   RH026             .551
    CH180     RH096            -.017   RH107            1.151
    CH180     1RH107             -1.   RH113             .004
    1CH180    RH001            1.151   2RH001            1.01
    1CH180    RH007            .0022   RH008            .0088
    1CH180    1RH008            .002   RH010             .065
    1CH180    RH016             .157   RH018              .03
    1CH180    RH019              .09   RH020             .083
    1CH180    2RH025            .005   RH026             .551
    1CH180    RH096            -.017   RH113             .004
    CH183     RH001              1.1   RH005             1.01
    CH183     RH007            .0012   RH008            .0018
    CH183     1RH008           .0003   RH010             .025
    CH183     RH015             .138   RH018             .037
    CH183     RH019             .161   1RH021            .178
    CH183     2RH025            .014   1RH026            .441
    CH183     RH096            -.017   RH107              1.1
    CH183     1RH107              1.   RH113            .0007
    CH188     1RH008             -1.   RH113               1.
    CH189     RH096             1.16   RH113              -1.
    CH190     RH096             1.16   RH114              -1.
    CH196     RH034              -1.   1RH034              1.
    CH196     RH063             -.94   2RH106              1.
    CH197     RH035              -1.   1RH035              1.
    CH197     RH067            1.745   2RH106              1.
    CH200     RH036              -1.   1RH036              1.
    CH200     RH072             7.31   2RH106              1.
    CH300     RH052              -1.   4RH052              1.
    CH300     1RH093             -5.   RH094              11.
    1CH302    R0000              .15   RH007             .014
    1CH302    RH008             .007   RH009             .039
    1CH302    RH012              .15   RH013              .14
    1CH302    1RH024            .215   2RH025             .27
    1CH302    1RH029             -1.   RH061            1.138
    1CH302    RH096            -.047   RH114              .08
    CH303     2RH023             -1.   RH046               1.
    CH303     1RH089           2.457   2RH089            -17.
    CH303     3RH089           -3.04
    CH304     1RH024             -1.   RH046               1.
    CH304     1RH089            -5.8   2RH089             34.
    CH304     3RH089         -11.422
    CH305     2RH028             -1.   RH046               1.
    CH305     1RH089            -.85   2RH089            -16.
    CH305     3RH089           2.934
    CH306     2RH023             -1.   RH043               1.
    CH306     RH080            -1.95   RH081            23.69
    CH306     RH082             -9.2   RH083             -18.
    CH306     RH084           -3.486
    CH307     2RH023             -1.   RH048               1.
    CH307     RH085             .295   RH086             2.43
    CH307     RH087             -9.2   RH088             -18.
    CH307     RH089           -3.486
    CH310     1RH024             -1.   RH048               1.
    CH310     RH085            -7.96   RH086            -6.23
    CH310     RH087             -8.3   RH088              33.
    CH310     RH089           -2.733
    CH311     1RH024             -1.   RH043               1.
    CH311     RH080           -10.21   RH081           12.896
    CH311     RH082             -8.3   RH083              33.
    CH311     RH084           -2.733
    CH314     2RH028             -1.   RH048               1.
    CH314     RH085            -3.01   RH086            8.065
    CH314     RH087              -8.   RH088             -17.
    CH314     RH089           -3.281
    CH360     RH018            .0235   1RH019          -.4382
    CH360     RH024           -.5393   1RH025           -.046
    CH360     2RH043              1.
    CH361     RH018            .0235   1RH019          -.4842
    CH361     RH024           -.5393   2RH043              1.
    CH362     RH018            .0275   2RH018          -.5444
    CH362     RH021           -.5576   2RH043              1.
    CH362     RH048            .0745   RH085           -2.445
    CH362     RH087           -.0745   RH088            -.447
    CH363     RH018            .0275   3RH018          -.5444
    CH363     RH021           -.5576   2RH043              1.
    CH363     RH048            .0745   RH085           -2.445
    CH363     RH087           -.0745   RH088            -.447
    CH401     RH018              -1.   RH053               1.
    CH401     RH115           -32.68   RH116             31.2
    CH401     RH117             -78.   RH118             -38.
    CH401     1RH118             95.
    CH404     RH018              -1.   RH052               1.
    CH404     RH093           -36.73   1RH093            -25.
    CH404     RH094             -38.   RH110            36.06
    CH406     1RH018             -1.   RH053               1.
    CH406     RH115           -25.16   RH116            24.68
    CH406     RH117             -71.   RH118            -36.2
    CH406     1RH118             -5.
    CH408     1RH018             -1.   RH109               1.
    CH408     1RH109          -25.15   2RH109          -18.43
    CH409     1RH018             -1.   RH052               1.
    CH409     RH093           -29.21   1RH093            -32.
    CH409     RH094            -35.8   RH110            28.54
    CH410     RH048              -1.   RH051               1.
    CH410     RH085             1.88   RH086           -5.609
    CH410     RH087              -1.   RH088               2.
    CH410     RH090           -16.58   1RH090            -16.
    CH410     2RH090           15.25   RH091             -18.
    CH410     RH103               1.   RH108             -30.
    CH411     RH048              -1.   RH053               1.
    CH411     RH085             1.88   RH086           -5.609
    CH411     RH087              -1.   RH088               2.
    CH411     RH115           -20.38   RH116             18.9
    CH411     RH117             -48.   RH118              28.
    CH411     1RH118             -5.
    CH413     RH048              -1.   RH085             1.88
    CH413     RH086           -5.609   RH087              -1.
    CH413     RH088               2.   RH109               1.
    CH413     1RH109          -20.38   2RH109            18.9
    CH414     RH048              -1.   RH052               1.
    CH414     RH085             1.88   RH086           -5.609
    CH414     RH087              -1.   RH088               2.
    CH414     RH093           -24.43   1RH093            -53.
    CH414     RH094             -28.   RH110            23.76
    CH415     2RH025             -1.   RH053               1.
    CH415     RH115            -9.11   RH116             7.63
    CH415     RH117             -10.   RH118             -10.
    CH415     1RH118             -5.
    CH417     2RH025             -1.   RH052               1.
    CH417     RH093           -13.16   1RH093            -15.
    CH417     RH094             -10.   RH110            12.49
    CH419     RH026              -1.   RH053               1.
    CH419     RH115             2.54   RH116            -4.02
    CH419     RH117              -6.   RH118              -.9
    CH419     1RH118             -5.
    CH421     RH026              -1.   RH052               1.
    CH421     RH093            -1.51   1RH093            -11.
    CH421     RH094              -.9   RH110              .84
    CH423     RH027              -1.   RH053               1.
    CH423     RH115              .55   RH116            -2.03
    CH423     RH117               1.   RH118             -3.9
    CH423     1RH118             -5.
    CH425     RH027              -1.   RH052               1.
    CH425     RH093             -3.5   1RH093             -4.
    CH425     RH094             -3.9   RH110             2.83
    CH427     2RH017             -1.   RH053               1.
    CH427     RH115            -1.82   RH116              .34
    CH427     RH117              11.   RH118             -14.
    CH427     1RH118             -5.
    CH430     2RH017             -1.   RH052               1.
    CH430     RH093            -5.87   1RH093              6.
    CH430     RH094             -14.   RH110              5.2
    CH432     RH017              -1.   RH053               1.
    CH432     RH115              .86   RH116            -2.34
    CH432     RH117               1.   RH118             -16.
    CH432     1RH118             -5.
    CH434     RH017              -1.   RH052               1.
    CH434     RH093            -3.19   1RH093             -4.
    CH434     RH094             -16.   RH110             2.52
    CH435     1RH017             -1.   RH051               1.
    CH435     RH090              8.9   1RH090             16.
    CH435     2RH090          -10.15   RH091             -4.5
    CH435     RH103               1.   RH108             -30.
    CH437     1RH017             -1.   RH109               1.
    CH437     1RH109            5.02   2RH109            -6.5
    CH438     1RH017             -1.   RH052               1.
    CH438     RH093              .97   1RH093            -21.
    CH438     RH094            -14.5   RH110            -1.64
    CH442     2RH026             -1.   RH053               1.
    CH442     RH115             2.57   RH116            -4.05
    CH442     RH117              34.   RH118            -35.5
    CH442     1RH118             -5.
    CH444     2RH026             -1.   RH052               1.
    CH444     RH093            -1.48   1RH093             29.
    CH444     RH094            -35.5   RH110              .81
    CH445     1RH026             -1.   RH053               1.
    CH445     RH115            -2.03   RH116              .55
    CH445     RH117              57.   RH118            -35.5
    CH445     1RH118             -5.
    CH447     1RH026             -1.   RH052               1.
    CH447     RH093            -6.08   1RH093             52.
    CH447     RH094            -35.5   RH110             5.41
    CH448     1RH027             -1.   RH053               1.
    CH448     RH115              .52   RH116              -2.
    CH448     RH117              71.   RH118            -36.7
    CH448     1RH118             -5.
    CH450     1RH027             -1.   RH052               1.
    CH450     RH093            -3.53   1RH093             66.
    CH450     RH094            -36.7   RH110             2.86
    CH451     RH030              -1.   RH051               1.
    CH451     RH090            17.72   1RH090              3.
    CH451     2RH090          -19.05   RH091              23.
    CH451     RH103               1.   RH108             -30.
    CH452     RH030              -1.   RH053               1.
    CH452     RH115            13.92   RH116            -15.4
    CH452     RH117             -29.   RH118              13.
    CH452     1RH118             -5.
    CH454     RH030              -1.   RH052               1.
    CH454     RH093             9.87   1RH093            -34.
    CH454     RH094              13.   RH110           -10.54
    CH455     3RH030             -1.   RH051               1.
    CH455     RH090            13.23   2RH090          -14.56
    CH455     RH091               6.   RH103               1.
    CH455     RH108             -30.
    CH456     3RH030             -1.   RH053               1.
    CH456     RH115             9.43   RH116           -10.91
    CH456     RH117             -32.   RH118              -4.
    CH456     1RH118             -5.
    CH459     3RH030             -1.   RH052               1.
    CH459     RH093             5.38   1RH093            -37.
    CH459     RH094              -4.   RH110            -6.05
    CH460     2RH030             -1.   RH051               1.
    CH460     RH090            16.41   1RH090             11.
    CH460     2RH090          -17.74   RH091              2.6
    CH460     RH103               1.   RH108             -30.
    CH461     2RH030             -1.   RH053               1.
    CH461     RH115            12.61   RH116           -14.09
    CH461     RH117             -21.   RH118             -7.4
    CH461     1RH118             -5.
    CH464     2RH030             -1.   RH052               1.
    CH464     RH093             8.56   1RH093            -26.
    CH464     RH094             -7.4   RH110            -9.23
    CH465     1RH030             -1.   RH051               1.
    CH465     RH090            18.45   1RH090              2.
    CH465     2RH090          -19.79   RH103               1.
    CH465     RH108             -30.
    CH467     1RH030             -1.   RH109               1.
    CH467     1RH109           14.66   2RH109          -16.14
    CH468     1RH030             -1.   RH052               1.
    CH468     RH093            10.61   1RH093            -35.
    CH468     RH094             -10.   RH110           -11.28
    CH472     1RH033             -1.   RH053               1.
    CH472     RH115             9.82   RH116            -11.3
    CH472     RH117              51.   RH118            -34.2
    CH472     1RH118             -5.
    CH474     1RH033             -1.   RH052               1.
    CH474     RH093             5.77   1RH093             46.
    CH474     RH094            -34.2   RH110            -6.44
    CH475     2RH033             -1.   RH053               1.
    CH475     RH115             7.32   RH116             -8.8
    CH475     RH117              59.   RH118            -34.5
    CH475     1RH118             -5.
    CH477     2RH033             -1.   RH052               1.
    CH477     RH093             3.27   1RH093             54.
    CH477     RH094            -34.5   RH110            -3.94
    CH478     4RH030             -1.   RH053               1.
    CH478     RH115            11.19   RH116           -12.67
    CH478     RH117              71.   RH118             -35.
    CH478     1RH118             -5.
    CH480     4RH030             -1.   RH052               1.
    CH480     RH093             7.14   1RH093             66.
    CH480     RH094             -35.   RH110            -7.81
    CH481     RH022             .934   RH030              -1.
    CH481     RH096             .032   RH099               1.
    CH481     RH113             .016
    CH482     RH022             .467   RH030              -1.
    CH482     RH051             .467   RH090           10.409
    CH482     1RH090           3.736   2RH090          -6.361
    CH482     RH091            10.04   RH096             .032
    CH482     RH099               1.   RH103             .467
    CH482     RH108            32.69   RH113             .016
    CH483     RH022              -1.   RH053               1.
    CH483     RH115             8.49   RH116            -9.97
    CH483     RH117             -24.   RH118             11.5
    CH483     1RH118             -5.
    CH485     RH022              -1.   RH052               1.
    CH485     RH093             4.44   1RH093            -29.
    CH485     RH094             11.5   RH110            -5.11
    CH486     1RH022            .934   3RH030             -1.
    CH486     RH096             .032   RH099               1.
    CH486     RH113             .016
    CH487     1RH022            .467   3RH030             -1.
    CH487     RH051             .467   RH090            5.193
    CH487     1RH090           1.401   2RH090          -5.814
    CH487     RH091            3.876   RH096             .032
    CH487     RH099               1.   RH103             .467
    CH487     RH108            32.69   RH113             .016
    CH488     1RH022             -1.   RH053               1.
    CH488     RH115             7.32   RH116             -8.8
    CH488     RH117             -29.   RH118             -1.7
    CH488     1RH118             -5.
    CH491     1RH022             -1.   RH052               1.
    CH491     RH093             3.27   1RH093            -34.
    CH491     RH094             -1.7   RH110            -3.94
    CH492     2RH022            .934   2RH030             -1.
    CH492     RH096             .032   RH099               1.
    CH492     RH113             .016
    CH493     2RH022            .467   2RH030             -1.
    CH493     RH051             .467   RH090           10.656
    CH493     1RH090           6.071   2RH090          -6.608
    CH493     RH091            2.148   RH096             .032
    CH493     RH099               1.   RH103             .467
    CH493     RH108            32.69   RH113             .016
    CH494     2RH022             -1.   RH053               1.
    CH494     RH115             9.02   RH116           -10.05
    CH494     RH117             -19.   RH118             -5.4
    CH494     1RH118             -5.
    CH497     2RH022             -1.   RH052               1.
    CH497     RH093             4.97   1RH093            -24.
    CH497     RH094             -5.4   RH110            -5.64
    CH498     3RH022            .934   1RH030             -1.
    CH498     RH096             .032   RH099               1.
    CH498     RH113             .016
    CH499     3RH022            .467   1RH030             -1.
    CH499     RH051             .467   RH090           10.676
    CH499     1RH090           1.401   2RH090          -6.627
    CH499     RH091             .934   RH096             .032
    CH499     RH099               1.   RH103             .467
    CH499     RH108            32.69   RH113             .016
    CH501     3RH022             -1.   RH109               1.
    CH501     1RH109            9.06   2RH109          -10.54
    CH502     3RH022             -1.   RH052               1.
    CH502     RH093             5.03   1RH093            -34.
    CH502     RH094              -8.   RH110            -5.68
    CH503     RH029              -1.   RH053               1.
    CH503     RH115            -5.18   RH116              3.7
    CH503     RH117              48.   RH118             -16.
    CH503     1RH118             -5.
    CH505     RH029              -1.   RH052               1.
    CH505     RH093            -9.23   1RH093             43.
    CH505     RH094             -16.   RH110             8.56
    CH506     1RH029             -1.   RH053               1.
    CH506     RH115            -6.03   RH116             4.55
    CH506     RH117              91.   RH118            -37.6
    CH506     1RH118             -5.
    CH508     1RH029             -1.   RH052               1.
    CH508     RH093           -10.08   1RH093             86.
    CH508     RH094            -37.7   RH110             9.41
    CH509     RH020              .09   RH029              -1.
    CH509     RH053              .78   RH096             .076
    CH509     RH099               1.   RH113              .04
    CH509     RH115            -6.29   RH116               5.
    CH509     RH117             5.22   RH118           -11.31
    CH509     1RH118           -4.35
    CH511     RH020              .09   RH029              -1.
    CH511     RH052              .78   RH093            -9.81
    CH511     1RH093             .87   RH094           -11.31
    CH511     RH096             .076   RH099               1.
    CH511     RH110             9.23   RH113              .04
    CH512     2RH023             .09   1RH029             -1.
    CH512     RH053              .78   RH096             .076
    CH512     RH099               1.   RH113              .04
    CH512     RH115            -8.74   RH116             8.33
    CH512     RH117             31.3   RH118           -32.45
    CH512     1RH118           -4.35
    CH514     2RH023             .09   1RH029             -1.
    CH514     RH052              .78   RH093           -12.27
    CH514     1RH093             27.   RH094           -32.45
    CH514     RH096             .076   RH099               1.
    CH514     RH110            11.68   RH113              .04
    CH515     4RH022             -1.   RH053               1.
    CH515     RH115            10.17   RH116           -11.65
    CH515     RH117             -12.   RH118              15.
    CH515     1RH118             -5.
    CH517     4RH022             -1.   RH052               1.
    CH517     RH093             6.12   1RH093            -17.
    CH517     RH094              15.   RH110            -6.79
    CH518     RH001               1.   1RH001            1.01
    CH518     RH007            .0024   RH008            .0099
    CH518     1RH008           .0022   RH010            .0382
    CH518     RH016            .1081   RH018            .0242
    CH518     RH019             .101   RH020            .0929
    CH518     2RH025            .005   RH026             .394
    CH518     1RH028           .0151   RH029            .0988
    CH518     RH030            .1014   1RH060           .2153
    CH518     RH096           -.0228   RH107               1.
    CH518     1RH107             -1.   RH113            .0048
    CH519     RH001            1.151   2RH001            1.01
    CH519     RH007            .0022   RH008            .0088
    CH519     1RH008            .002   RH010             .065
    CH519     RH016             .157   RH018              .03
    CH519     RH019              .09   RH020             .083
    CH519     2RH025            .007   RH026            .3019
    CH519     1RH028           .0174   RH029            .1143
    CH519     RH030            .1174   1RH060           .2491
    CH519     RH096           -.0237   RH107            1.151
    CH519     1RH107             -1.   RH113             .004
    CH520     RH001            1.027   RH003             1.01
    CH520     RH007            .0039   RH008            .0092
    CH520     1RH008           .0004   RH010            .0485
    CH520     RH016            .1344   RH018            .0394
    CH520     RH019            .1373   RH020             .107
    CH520     2RH025            .005   RH027            .2999
    CH520     1RH028           .0251   RH029            .1168
    CH520     RH030            .0805   1RH060           .2224
    CH520     RH096            -.023   RH107            1.027
    CH520     1RH107             -1.   RH113            .0006
    CH521     RH001              1.1   RH006             1.01
    CH521     RH007             .002   RH008            .0072
    CH521     1RH008           .0005   RH010            .0378
    CH521     RH016            .1495   2RH017           .2627
    CH521     RH018            .0436   3RH018           .1228
    CH521     RH021            .1257   2RH025            .005
    CH521     1RH028           .0191   RH029            .1279
    CH521     3RH030           .0912   1RH060           .2382
    CH521     RH096           -.0234   RH107              1.1
    CH521     1RH107             -1.   RH113             .003
    CH522     RH001              1.1   RH002             1.01
    CH522     RH007            .0024   RH008             .009
    CH522     1RH008           .0019   RH010            .0259
    CH522     RH015            .1192   RH017            .2843
    CH522     RH018            .0354   2RH018           .1273
    CH522     RH021            .1333   2RH025            .005
    CH522     1RH028           .0198   RH029            .1327
    CH522     2RH030           .0956   1RH060           .2481
    CH522     RH096           -.0237   RH107              1.1
    CH522     1RH107             -1.   RH113            .0062
    CH524     RH001              1.1   1RH005            1.01
    CH524     RH007            .0002   RH008            .0004
    CH524     1RH008           .0003   RH010            .0236
    CH524     RH015              .07   RH018              .03
    CH524     RH019              .15   RH023              .22
    CH524     2RH025            .005   2RH026           .2619
    CH524     1RH028            .019   1RH029           .0834
    CH524     1RH033           .1357   1RH060           .2381
    CH524     RH096           -.0234   RH107              1.1
    CH524     1RH107             -1.   RH113            .0005
    CH525     RH001              1.1   RH005             1.01
    CH525     RH007            .0012   RH008            .0018
    CH525     1RH008           .0003   RH010             .025
    CH525     RH015             .138   RH018             .037
    CH525     RH019             .161   1RH021            .178
    CH525     2RH025            .014   1RH026           .2032
    CH525     1RH028            .026   1RH029           .1424
    CH525     2RH033            .069   1RH060           .2378
    CH525     RH096           -.0234   RH107              1.1
    CH525     1RH107              1.   RH113            .0007
    CHF28     RH051              -1.   1RH051              1.
    CHF28     RH091               3.
    1CHF29    2RH051              1.   RH052              -1.
    1CHF29    1RH093             -3.   RH094               5.
    CHF35     RH052              -1.   3RH052              1.
    CHF35     RH093             3.08   RH094               5.
    CHF35     RH110            -3.19
    CHF48     RH053              -1.   3RH053              1.
    CHF48     RH117              -3.   RH118               5.
    CHF80     RH052              -1.   2RH052              1.
    CHF80     RH094               5.
    CHF82     RH053              -1.   2RH053              1.
    CHF82     RH115            -1.05   RH116             1.67
    CHF82     RH117              -5.   RH118              15.
    C1002     R1002             1.01   R1008            .0005
    C1002     R1012             .012   R1013            .0775
    C1002     R1015              .09   R1016             .031
    C1002     R1017             .026   R1018             .134
    C1002     R1019             .141   R1022             .051
    C1002     R1025             .437   R1039            -.026
    C1002     R1071            1.028   1R1071            .973
    C1002     2R1071              1.
    C1003     R1003             1.01   R1008            .0025
    C1003     R1010             .003   R1012             .011
    C1003     R1013            .0635   R1015             .064
    C1003     R1016             .022   R1017             .022
    C1003     R1018             .103   R1019             .118
    C1003     R1022             .044   R1026             .547
    C1003     R1039           -.0212   R1071               1.
    C1003     1R1071              1.   2R1071              1.
    C1004     R1004             1.01   R1008            .0025
    C1004     R1010            .0025   R1012              .01
    C1004     R1014             .065   R1015             .082
    C1004     R1016             .026   R1017             .026
    C1004     R1018             .127   R1020             .141
    C1004     R1023             .054   R1027             .464
    C1004     R1039           -.0247   R1071            1.018
    C1004     1R1071            .973   2R1071              1.
    C1005     R1005             1.01   R1008            .0016
    C1005     R1010            .0022   R1012             .011
    C1005     R1014            .0642   R1015             .081
    C1005     R1016              .02   R1017             .025
    C1005     R1018             .112   1R1019            .127
    C1005     R1022             .046   R1028              .51
    C1005     R1039           -.0212   R1071             .999
    C1005     1R1071              1.   2R1071              1.
    1C1005    R1005             .808   1R1006            .202
    1C1005    2R1006            .202   R1008            .0017
    1C1005    R1010             .002   R1012            .0104
    1C1005    R1014            .0621   R1015            .0778
    1C1005    R1016             .021   R1017             .025
    1C1005    R1018            .1096   3R1021           .1256
    1C1005    R1023            .0498   1R1027            .515
    1C1005    R1039           -.0213   R1071            1.004
    1C1005    1R1071          1.0032   2R1071              1.
    1C1006    1R1007            1.01   R1008            .0035
    1C1006    R1010            .0005   R1012             .007
    1C1006    R1013             .067   R1015             .107
    1C1006    R1016             .032   R1017             .033
    1C1006    R1018              .17   1R1020            .158
    1C1006    R1023             .057   1R1028            .365
    1C1006    R1039           -.0247   R1071            1.052
    1C1006    1R1071             .95   2R1071              1.
    C1007     R1007             1.01   R1008             .001
    C1007     R1012             .004   R1013             .044
    C1007     R1015             .093   R1016             .027
    C1007     R1017              .03   R1018             .141
    C1007     1R1021            .193   R1024             .069
    C1007     R1029             .398   R1039           -.0264
    C1007     R1071            1.024   1R1071            .992
    C1007     2R1071              1.
    C1008     1R1004            1.01   R1008             .001
    C1008     R1010             .001   R1012             .002
    C1008     R1013             .037   R1015             .048
    C1008     R1016              .02   R1017              .02
    C1008     R1018             .138   2R1021            .266
    C1008     R1024             .031   2R1028            .436
    C1008     R1039           -.0264   R1071             .973
    C1008     1R1071            .992   2R1071              1.
    1C1008    1R1004            .808   1R1006            .202
    1C1008    2R1006            .202   R1008            .0012
    1C1008    R1010             .001   R1012            .0032
    1C1008    R1013            .0404   R1015            .0514
    1C1008    R1016             .021   R1017             .021
    1C1008    R1018            .1304   R1021            .2368
    1C1008    R1024            .0378   2R1028           .4558
    1C1008    R1039           -.0254   R1071            .9834
    1C1008    1R1071           .9968   2R1071              1.
    C1009     R1002            .7575   1R1006           .2525
    C1009     2R1006           .2525   R1008            .0009
    C1009     R1010            .0003   R1012             .011
    C1009     R1013            .0716   R1015            .0838
    C1009     R1016            .0296   R1017            .0258
    C1009     R1018            .1255   3R1021           .1357
    C1009     R1023            .0546   3R1028           .4612
    C1009     R1039           -.0248   R1071           1.0273
    C1009     1R1071           .9838   2R1071              1.
    2C1009    R1003            .7575   1R1006           .2525
    2C1009    2R1006           .2525   R1008            .0024
    2C1009    R1010            .0026   R1012            .0103
    2C1009    R1013            .0611   R1015            .0643
    2C1009    R1016            .0228   R1017            .0228
    2C1009    R1018            .1022   3R1021           .1185
    2C1009    R1023            .0493   3R1028           .5437
    2C1009    R1039           -.0212   R1071           1.0063
    2C1009    1R1071           1.004   2R1071              1.
    C1012     R1008              -1.   R1039             1.19
    C1013     R1009              -1.   R1039             1.19
    1C1013    R0000             -16.   R1009              -1.
    1C1013    1R1009              1.
    C1015     R0000             16.7   R1010               1.
    1C1015    R1009               1.   R1010              -1.
    C1016     R0000            -11.5   1R1009              1.
    C1016     R1012              -1.
    C1017     R1008               1.   R1010              -1.
    C1018     R0000              12.   R1012               1.
    C1020     R1008               1.   R1012              -1.
    C1021     R0000             .228   R1012              -1.
    C1021     R1046               1.   R1075            -6.38
    C1021     R1076            -.579   R1077       -91.399899
    C1021     R1078       225.899897
    C1022     R0000             .114   R1012              -1.
    C1022     R1050               1.   R1079            -23.8
    C1022     R1080            -.772   R1081       -91.399899
    C1022     R1082       225.899897
    C1023     R1014              -1.   R1039             1.11
    C1023     1R1039              1.   R1042              -1.
    1C1023    R1013              -1.   R1014               1.
    C1024     R0000             .196   R1014              -1.
    C1024     R1046               1.   R1075            18.22
    C1024     R1076            -.498   R1077            -63.7
    C1024     R1078             28.1
    C1025     R0000             .098   R1014              -1.
    C1025     R1050               1.   R1079             9.33
    C1025     R1080            -.664   R1081            -63.7
    C1025     R1082             28.1
    C1026     R1014              -1.   R1055               1.
    C1026     UKLDF              -1.
    1C1027    R1013              -1.   R1054               1.
    1C1027    R1070               1.   UKLDF              -1.
    C1028     R1015              -1.   R1039             1.11
    C1028     1R1039              1.   R1042              -1.
    C1029     R0000              .09   R1015              -1.
    C1029     R1050               1.   R1079             22.2
    C1029     R1080             -.61   R1081              30.
    C1029     R1082            -36.8
    C1030     R1015              -1.   R1055               1.
    C1030     UKLDF              -1.
    C1032     R1016              -1.   R1055               1.
    C1032     UKLDF              -1.
    C1034     R1016              -1.   R1062               1.
    C1034     R1091            10.58   R1092             -8.5
    C1034     R1093           -21.48   R1094             56.8
    C1034     R1095              -7.   R1096            -100.
    C1035     R1016              -1.   R1063               1.
    C1035     R1097            13.41   R1098            -13.5
    C1035     R1099           -30.52   R1100             56.8
    C1035     R1101            14.88
    C1036     R1016              -1.   R1065               1.
    C1036     R1106           -36.06   R1107            35.35
    C1036     R1108            -18.5   R1109             56.8
    C1036     R1110            -117.
    C1037     R1016              -1.   R1068               1.
    C1037     R1111           -35.54   R1112            34.85
    C1037     R1113            -38.5   R1114             56.8
    C1037     R1115            -104.
    C1038     R1016             -.52   R1017             -.48
    C1038     R1055               1.   UKLDF              -1.
    C1040     R1016             -.52   R1017             -.48
    C1040     R1062               1.   R1091             8.89
    C1040     R1092             -8.5   R1093           -21.16
    C1040     R1094            38.71   R1095              -7.
    C1040     R1096            -100.
    C1041     R1016             -.52   R1017             -.48
    C1041     R1063               1.   R1097            11.72
    C1041     R1098            -13.5   R1099           -30.06
    C1041     R1100            38.71   R1101            14.88
    C1042     R1016             -.52   R1017             -.48
    C1042     R1065               1.   R1106           -36.06
    C1042     R1107            35.35   R1108            -18.5
    C1042     R1109            38.71   R1110            -117.
    C1043     R1016             -.52   R1017             -.48
    C1043     R1068               1.   R1111           -35.54
    C1043     R1112            34.85   R1113            -38.5
    C1043     R1114            38.71   R1115            -104.
    1C1043    R1016             -.52   R1017             -.48
    1C1043    R1039             1.09   1R1039              1.
    1C1043    R1042              -1.
    C1044     R1016             -.26   R1017             -.24
    C1044     R1018              -.5   R1057               1.
    C1044     1R1057              1.
    C1045     R1017              -.6   R1018              -.4
    C1045     R1057               1.   1R1057              1.
    C1047     R1017            -.185   R1018            -.575
    C1047     R1039           -.0194   R1058              .76
    C1047     R1074               1.
    C1048     R1017             -.16   R1018             -.84
    C1048     R1060               1.   R1086             1.96
    C1048     R1087              -7.   R1088            -5.78
    C1048     R1089             6.62   R1090              -7.
    C1049     R1017             -.16   R1018             -.84
    C1049     R1062               1.   R1091             3.28
    C1049     R1092              -7.   R1093           -18.67
    C1049     R1094             6.62   R1095              -7.
    C1049     R1096            -100.
    C1050     R1017             -.16   R1018             -.84
    C1050     R1063               1.   R1097             6.11
    C1050     R1098             -12.   R1099           -27.23
    C1050     R1100             6.62   R1101             9.08
    C1052     R1017             -.16   R1018             -.84
    C1052     R1065               1.   R1106           -30.26
    C1052     R1107            29.55   R1108             -17.
    C1052     R1109             6.62   R1110             -99.
    C1053     R1017             -.16   R1018             -.84
    C1053     R1068               1.   R1111           -29.74
    C1053     R1112            29.05   R1113             -37.
    C1053     R1114             6.62   R1115             -86.
    C1055     R1018              -1.   R1060               1.
    C1055     R1086              1.2   R1087             -6.6
    C1055     R1088            -5.41   R1089              .58
    C1055     R1090              -7.
    C1056     R1018              -1.   R1062               1.
    C1056     R1091             2.52   R1092             -6.6
    C1056     R1093            -18.2   R1094              .58
    C1056     R1095              -7.   R1096            -100.
    C1057     R1018              -1.   R1063               1.
    C1057     R1097             5.35   R1098            -11.6
    C1057     R1099           -26.69   R1100              .58
    C1057     R1101             7.98
    C1058     R1018              -1.   R1064               1.
    C1058     R1102           -21.98   1R1102           20.84
    C1058     R1103            -28.6   R1104              .58
    C1058     R1105             -47.
    C1059     R1018              -1.   R1065               1.
    C1059     R1106           -29.16   R1107            28.45
    C1059     R1108            -16.6   R1109              .58
    C1059     R1110             -99.
    1C1059    R1018              -1.   R1039             1.08
    C1060     R1018              -1.   R1068               1.
    C1060     R1111           -28.64   R1112            27.95
    C1060     R1113            -36.6   R1114              .58
    C1060     R1115             -86.
    1C1060    R1019               1.   1R1019             -1.
    2C1060    R1021               1.   1R1021             -1.
    C1061     R1019              -1.   R1060               1.
    C1061     R1086            -1.46   R1087              1.2
    C1061     R1088            13.32   R1089            -2.94
    C1061     R1090              -5.
    1C1061    R1020               1.   1R1020             -1.
    2C1061    R1021               1.   2R1021             -1.
    C1062     R1019              -1.   R1062               1.
    C1062     R1091             -.14   R1092              1.2
    C1062     R1093              .84   R1094            -2.94
    C1062     R1095              -5.
    C1063     R1019              -1.   R1063               1.
    C1063     R1097             2.69   R1098             -3.8
    C1063     R1099            -7.44   R1100            -2.94
    C1063     R1101             2.33
    C1064     R1019              -1.   R1064               1.
    C1064     R1102           -16.33   1R1102           15.19
    C1064     R1103            -20.8   R1104            -2.94
    C1064     R1105             -18.
    C1065     R1019              -1.   R1068               1.
    C1065     R1111           -22.99   R1112             22.3
    C1065     R1113            -28.8   R1114            -2.94
    C1065     R1115             -57.
    1C1065    R1019              -1.   R1039             1.06
    C1066     R1020              -1.   R1060               1.
    C1066     R1086            -3.32   R1087             -2.5
    C1066     R1088             13.1   R1089            -2.94
    C1066     R1090              -3.
    C1067     R1020              -1.   R1062               1.
    C1067     R1091              -2.   R1092             -2.5
    C1067     R1093              .83   R1094            -2.94
    C1067     R1095              -3.
    C1068     R1020              -1.   R1063               1.
    C1068     R1097              .83   R1098             -7.5
    C1068     R1099            -7.32   R1100            -2.94
    C1068     R1101              .37
    C1069     R1020              -1.   R1064               1.
    C1069     R1102           -14.37   1R1102           13.23
    C1069     R1103            -24.5   R1104            -2.94
    C1069     R1105             -18.
    C1070     R1021              -1.   R1060               1.
    C1070     R1086            -4.84   R1087             -8.3
    C1070     R1088            12.99   R1089            -2.94
    C1070     R1090               7.
    1C1070    3R1021             -1.   R1060               1.
    1C1070    R1086            -1.17   R1087              -1.
    1C1070    R1088            13.43   R1089            -2.94
    1C1070    R1090              -7.
    C1071     R1021              -1.   R1062               1.
    C1071     R1091            -3.52   R1092             -8.3
    C1071     R1093              .82   R1094            -2.94
    C1071     R1095               7.
    1C1071    3R1021             -1.   R1062               1.
    1C1071    R1091             -.15   R1092              -1.
    1C1071    R1093              .85   R1094            -2.94
    1C1071    R1095              -7.
    C1072     R1021              -1.   R1063               1.
    C1072     R1097             -.69   R1098            -13.3
    C1072     R1099            -7.25   R1100            -2.94
    C1072     R1101              .78
    C1074     R1021              -1.   R1065               1.
    C1074     R1106           -21.96   R1107            21.25
    C1074     R1108            -18.3   R1109            -2.94
    C1074     R1110             -70.
    C1076     R1019             -.86   R1022             -.14
    C1076     R1064               1.   R1102           -13.57
    C1076     1R1102           12.43   R1103             -19.
    C1076     R1104            -3.17   R1105             -17.
    C1078     R1020             -.88   R1023             -.12
    C1078     R1064               1.   R1102           -14.48
    C1078     1R1102           13.34   R1103            -23.7
    C1078     R1104            -3.17   R1105             -17.
    C1081     R1022              -1.   R1062               1.
    C1081     R1091            -5.54   R1092              7.8
    C1081     R1093       113.399899   R1094            -3.47
    C1081     R1095               1.
    C1082     R1022              -1.   R1063               1.
    C1082     R1097            -2.71   R1098              2.8
    C1082     R1099       105.459998   R1100            -3.47
    C1082     R1101            -3.64
    C1083     R1022              -1.   R1065               1.
    C1083     R1106           -17.54   R1107            16.83
    C1083     R1108             -2.2   R1109            -3.47
    C1083     R1110             -37.
    C1084     R1022              -1.   R1068               1.
    C1084     R1111           -17.02   R1112            16.33
    C1084     R1113            -22.2   R1114            -3.47
    C1084     R1115             -24.
    1C1084    R1022              -1.   R1039             1.05
    C1085     R1023              -1.   R1062               1.
    C1085     R1091            -5.54   R1092              4.1
    C1085     R1093       113.399899   R1094            -3.47
    C1085     R1095               2.
    1C1085    R1023              -1.   R1063               1.
    1C1085    R1097            -2.71   R1098              -.9
    1C1085    R1099       105.459998   R1100            -3.47
    1C1085    R1101            -5.02
    C1086     R1023              -1.   R1068               1.
    C1086     R1111           -15.64   R1112            14.95
    C1086     R1113            -25.9   R1114            -3.47
    C1086     R1115             -24.
    C1087     R1023              -1.   R1065               1.
    C1087     R1106           -16.16   R1107            15.45
    C1087     R1108             -5.9   R1109            -3.47
    C1087     R1110             -37.
    C1088     R1024              -1.   R1062               1.
    C1088     R1091             -4.2   R1092             -7.4
    C1088     R1093         114.3799   R1094            -3.61
    C1088     R1095               7.
    C1089     R1024              -1.   R1063               1.
    C1089     R1097            -1.37   R1098            -12.4
    C1089     R1099       106.370098   R1100            -3.61
    C1089     R1101            -1.84
    C1090     R1024              -1.   R1065               1.
    C1090     R1106           -19.34   R1107            18.63
    C1090     R1108            -17.4   R1109            -3.61
    C1090     R1110             -37.
    C1091     R1024              -1.   R1068               1.
    C1091     R1111           -18.82   R1112            18.13
    C1091     R1113            -37.4   R1114            -3.61
    C1091     R1115             -24.
    1C1091    3R1028             -1.   R1030               .5
    1C1091    2R1033            .598   R1037            -.098
    1C1091    R1039           -.0154   R1072            1.005
    2C1091    3R1028             -1.   R1065               1.
    2C1091    R1106             -.26   R1107             -.45
    2C1091    R1108             10.5   R1109              -.8
    2C1091    R1110               6.
    3C1091    3R1028             -1.   R1068               1.
    3C1091    R1111              .26   R1112             -.95
    3C1091    R1113             -9.5   R1114              -.8
    3C1091    R1115              19.
    C1092     R1025              -1.   R1030              .52
    C1092     R1032               .6   R1037             -.12
    C1092     R1039           -.0154   R1072            1.003
    C1093     R1025              -1.   R1065               1.
    C1093     R1106             -.06   R1107             -.65
    C1093     R1108             20.5   R1109              -.8
    C1093     R1110             -10.
    C1094     R1025              -1.   R1068               1.
    C1094     R1111              .46   R1112            -1.15
    C1094     R1113               .5   R1114              -.8
    C1094     R1115               3.
    C1095     R1026              -1.   R1030             .479
    C1095     R1032             .618   R1037            -.097
    C1095     R1039           -.0154   R1072               1.
    C1096     R1026              -1.   R1065               1.
    C1096     R1106             1.24   R1107            -1.95
    C1096     R1108             21.5   R1109              -.8
    C1096     R1110             -20.
    C1097     R1026              -1.   R1068               1.
    C1097     R1111             1.76   R1112            -2.45
    C1097     R1113              1.5   R1114              -.8
    C1097     R1115              -7.
    1C1097    1R1027             -1.   R1030             .464
    1C1097    3R1033            .691   R1037            -.155
    1C1097    R1039           -.0154   R1072               1.
    2C1097    1R1027             -1.   R1065               1.
    2C1097    R1106             1.73   R1107            -2.44
    2C1097    R1108              1.3   R1109              -.8
    2C1097    R1110               2.
    C1098     R1027              -1.   R1030             .577
    C1098     R1033             .536   R1037            -.113
    C1098     R1039           -.0154   R1072            1.007
    C1099     R1027              -1.   R1065               1.
    C1099     R1106              .24   R1107             -.95
    C1099     R1108               3.   R1109              -.8
    C1099     R1110             -10.
    C1100     R1027              -1.   R1068               1.
    C1100     R1111              .76   R1112            -1.45
    C1100     R1113             -17.   R1114              -.8
    C1100     R1115               3.
    C1101     R1028              -1.   R1030             .454
    C1101     R1033             .732   R1037            -.186
    C1101     R1039           -.0154   R1072            1.044
    C1102     R1028              -1.   R1065               1.
    C1102     R1106             2.44   R1107            -3.15
    C1102     R1108              6.3   R1109              -.8
    C1102     R1110             -10.
    C1103     R1028              -1.   R1068               1.
    C1103     R1111             2.96   R1112            -3.65
    C1103     R1113            -13.7   R1114              -.8
    C1103     R1115               3.
    1C1103    1R1028             -1.   R1030             .653
    1C1103    1R1033            .347   R1039           -.0154
    1C1103    R1072            1.052
    2C1103    1R1028             -1.   R1065               1.
    2C1103    R1106            -5.35   R1107             4.64
    2C1103    R1108             -4.5   R1109              -.8
    2C1103    R1110               4.
    3C1103    1R1028             -1.   R1068               1.
    3C1103    R1111            -4.83   R1112             4.14
    3C1103    R1113            -24.5   R1114              -.8
    3C1103    R1115              17.
    C1104     R1021            -.037   R1024            -.008
    C1104     R1029              -1.   R1031              .77
    C1104     R1034             .275   R1039           -.0154
    C1104     R1072            1.246
    1C1104    R1021            -.073   R1024            -.016
    1C1104    2R1028             -1.   R1031              .63
    1C1104    R1034             .459   R1039           -.0154
    1C1104    R1072             1.02
    2C1104    2R1028             -1.   R1065               1.
    2C1104    R1106             -1.2   R1107              .49
    2C1104    R1108            -14.5   R1109              -.8
    2C1104    R1110              -3.
    3C1104    2R1028             -1.   R1068               1.
    3C1104    R1111             -.68   R1112             -.01
    3C1104    R1113            -34.5   R1114              -.8
    3C1104    R1115              10.
    C1105     R1029              -1.   R1065               1.
    C1105     R1106            -4.02   R1107             3.31
    C1105     R1108            -16.1   R1109              -.8
    C1105     R1110              50.
    C1106     R1029              -1.   R1068               1.
    C1106     R1111             -3.5   R1112             2.81
    C1106     R1113            -36.1   R1114              -.8
    C1106     R1115              63.
    C1108     R0000            .0504   R1009             .075
    C1108     R1010             .039   R1012             .095
    C1108     R1030              -1.   R1035              .16
    C1108     R1036             .202   R1037             .233
    C1108     1R1038            .015   R1039            .0437
    C1108     R1073              1.5
    1C1108    R0000            .0504   R1009             .076
    1C1108    R1010              .04   R1012             .097
    1C1108    R1022             -.15   R1030             -.85
    1C1108    R1035             .159   R1036             .205
    1C1108    R1037             .229   1R1038            .015
    1C1108    R1039            .0466   R1073              1.5
    C1109     R0000            .0504   R1009             .025
    C1109     R1010             .014   R1012             .054
    C1109     R1030              -1.   R1035             .111
    C1109     R1036             .192   R1037             .227
    C1109     1R1038            .246   R1039            .0552
    C1109     R1073               1.
    C1111     R0000            .0504   R1009             .067
    C1111     R1010             .036   R1012             .089
    C1111     R1030              -1.   R1035             .144
    C1111     R1036             .176   R1037             .194
    C1111     1R1038            .144   R1039            .0532
    C1111     R1073             1.11
    C1112     R0000            .0504   R1009             .049
    C1112     R1010             .026   R1012             .073
    C1112     R1030              -1.   R1035              .13
    C1112     R1036             .185   R1037             .209
    C1112     1R1038            .189   R1039             .055
    C1112     R1073               1.
    1C1112    R0000            .0504   R1009             .047
    1C1112    R1010             .027   R1012             .076
    1C1112    R1022             -.15   R1030             -.85
    1C1112    R1035             .128   R1036             .188
    1C1112    R1037             .203   1R1038            .191
    1C1112    R1039            .0557   R1073               1.
    C1113     R0000            .0504   R1009              .03
    C1113     R1010              .02   R1012             .053
    C1113     R1031              -1.   R1035             .101
    C1113     R1036             .188   R1038             .195
    C1113     2R1038            .281   R1039            .0509
    C1113     R1073               1.
    C1116     R0000            .0504   R1009             .017
    C1116     R1010             .013   R1012             .036
    C1116     R1031              -1.   R1035             .087
    C1116     R1036              .19   R1038             .208
    C1116     2R1038            .323   R1039            .0463
    C1116     R1073             1.11
    C1117     R0000            .0504   R1009             .047
    C1117     R1010             .029   R1012             .081
    C1117     R1031              -1.   R1035             .119
    C1117     R1036             .181   R1038             .177
    C1117     2R1038            .224   R1039            .0526
    C1117     R1073               1.
    1C1117    R0000            .0504   R1009             .048
    1C1117    R1010              .03   R1012             .082
    1C1117    R1024             -.17   R1031             -.83
    1C1117    R1035             .117   R1036             .183
    1C1117    R1038             .172   2R1038            .227
    1C1117    R1039            .0531   R1073               1.
    C1118     R0000            .0504   R1009             .058
    C1118     R1010             .035   R1012             .098
    C1118     R1031              -1.   R1035              .17
    C1118     R1036             .207   R1038             .212
    C1118     2R1038            .043   R1039            .0397
    C1118     R1073             1.67
    1C1118    R0000            .0504   R1009             .058
    1C1118    R1010             .036   R1012             .099
    1C1118    R1024             -.17   R1031             -.83
    1C1118    R1035             .169   R1036             .211
    1C1118    R1038             .208   2R1038            .044
    1C1118    R1039            .0416   R1073             1.67
    2C1118    3R1033             -1.   R1065               1.
    2C1118    R1106             3.01   R1107            -3.72
    2C1118    R1108              7.4   R1109              -.8
    2C1118    R1110             -19.
    3C1118    3R1033             -2.   R1068               1.
    3C1118    R1111             3.53   R1112            -4.22
    3C1118    R1113            -12.6   R1114              -.8
    3C1118    R1115              -6.
    C1119     R1032              -1.   R1064               1.
    C1119     R1102            10.19   1R1102          -11.33
    C1119     R1103             18.9   R1104             -.08
    C1119     R1105              -2.
    C1120     R1032              -1.   R1068               1.
    C1120     R1111             3.53   R1112            -4.22
    C1120     R1113             10.9   R1114              -.8
    C1120     R1115             -41.
    1C1120    R1032              -1.   R1065               1.
    1C1120    R1106             3.01   R1107            -3.72
    1C1120    R1108             30.9   R1109              -.8
    1C1120    R1110             -54.
    2C1120    2R1033             -1.   R1065               1.
    2C1120    R1106             3.01   R1107            -3.72
    2C1120    R1108             18.5   R1109              -.8
    2C1120    R1110             -17.
    3C1120    2R1033             -1.   R1068               1.
    3C1120    R1111             3.53   R1112            -4.22
    3C1120    R1113             -1.5   R1114              -.8
    3C1120    R1115              -4.
    C1121     R1033              -1.   R1064               1.
    C1121     R1102            10.19   1R1102          -11.33
    C1121     R1103              -.5   R1104             -.08
    C1121     R1105               2.
    C1122     R1033              -1.   R1065               1.
    C1122     R1106             3.01   R1107            -3.72
    C1122     R1108             11.5   R1109              -.8
    C1122     R1110             -50.
    C1123     R1033              -1.   R1068               1.
    C1123     R1111             3.53   R1112            -4.22
    C1123     R1113             -8.5   R1114              -.8
    C1123     R1115             -37.
    1C1123    1R1033             -1.   R1065               1.
    1C1123    R1106             3.09   R1107             -3.8
    1C1123    R1108              1.5   R1109              -.8
    1C1123    R1110             -17.
    2C1123    1R1033             -1.   R1068               1.
    2C1123    R1111             3.61   R1112             -4.3
    2C1123    R1113            -18.5   R1114              -.8
    2C1123    R1115              -4.
    3C1123    1R1033             -1.   R1064               1.
    3C1123    R1102            10.27   1R1102          -11.41
    3C1123    R1103            -10.5   R1104             -.08
    3C1123    R1105              35.
    C1124     R1034              -1.   R1065               1.
    C1124     R1106             3.01   R1107            -3.72
    C1124     R1108            -15.9   R1109              -.8
    C1124     R1110              20.
    C1125     R1034              -1.   R1068               1.
    C1125     R1111             3.53   R1112            -4.22
    C1125     R1113            -35.9   R1114              -.8
    C1125     R1115              33.
    C1128     R1017           -.0075   R1018             -.04
    C1128     R1022           -.0475   R1065            -.905
    C1128     R1066               1.   R1108            -5.88
    C1128     R1110             14.5
    C1130     R1039               1.   R1068              -1.
    C1131     R0000             .194   R1035              -1.
    C1131     R1046               1.   R1075              .73
    C1131     R1076            -.493   R1077       -76.399899
    C1131     R1078             43.5
    C1132     R0000             .097   R1035              -1.
    C1132     R1050               1.   R1079            -19.7
    C1132     R1080            -.658   R1081       -76.399899
    C1132     R1082             43.5
    C1133     R0000              .17   R1036            -1.27
    C1133     R1037              .27   R1046               1.
    C1133     R1075             2.05   R1076            -.431
    C1133     R1077             52.6   R1078            -50.1
    C1134     R0000             .085   R1036            -1.27
    C1134     R1037              .27   R1050               1.
    C1134     R1079            -15.1   R1080            -.574
    C1134     R1081             52.6   R1082            -50.1
    C1135     R0000             .165   R1036              -1.
    C1135     R1046               1.   R1075               2.
    C1135     R1076            -.419   R1077             52.4
    C1135     R1078            -48.8
    C1136     R0000             .082   R1036              -1.
    C1136     R1050               1.   R1079            -14.7
    C1136     R1080            -.559   R1081             52.4
    C1136     R1082            -48.8
    C1137     R1037              -1.   R1060               1.
    C1137     R1086           -13.81   R1087             17.3
    C1137     R1088             7.02   R1089            -3.48
    C1137     R1090              43.
    C1138     R1037              -1.   R1062               1.
    C1138     R1091           -12.49   R1092             17.3
    C1138     R1093            -4.21   R1094            -3.48
    C1138     R1095              43.
    C1139     R1037              -1.   R1063               1.
    C1139     R1097            -9.66   R1098             12.3
    C1139     R1099           -11.66   R1100            -3.48
    C1139     R1101             1.33
    C1140     R1037              -1.   R1064               1.
    C1140     R1102           -15.33   1R1102           14.19
    C1140     R1103             -3.5   R1104            -3.48
    C1140     R1105             -22.
    C1141     R1037              -1.   R1068               1.
    C1141     R1111           -21.99   R1112             21.3
    C1141     R1113            -11.5   R1114            -3.48
    C1141     R1115             -61.
    C1142     R1038              -1.   R1060               1.
    C1142     R1086           -12.51   R1087             -5.8
    C1142     R1088              7.5   R1089            -3.48
    C1142     R1090              40.
    C1143     R1038              -1.   R1062               1.
    C1143     R1091           -11.19   R1092             -5.8
    C1143     R1093            -4.25   R1094            -3.48
    C1143     R1095              40.
    C1144     R1038              -1.   R1064               1.
    C1144     R1102           -15.33   1R1102           14.19
    C1144     R1103            -27.5   R1104            -3.48
    C1144     R1105             -22.
    C1145     R1038              -1.   R1065               1.
    C1145     R1106           -22.51   R1107             21.8
    C1145     R1108            -15.5   R1109            -3.48
    C1145     R1110             -74.
    C1146     R1038              -1.   R1068               1.
    C1146     R1111           -21.99   R1112             21.3
    C1146     R1113            -35.5   R1114            -3.48
    C1146     R1115             -61.
    C1147     R1037               1.   R1038              -1.
    1C1147    1R1038             -1.   R1068               1.
    1C1147    R1111           -14.54   R1112            13.85
    1C1147    R1113            -11.5   R1114            -3.44
    1C1147    R1115             -10.
    2C1147    2R1038             -1.   R1068               1.
    2C1147    R1111           -14.54   R1112            13.85
    2C1147    R1113            -35.5   R1114            -3.44
    2C1147    R1115              17.
    C1148     1R1038             -1.   R1039               1.
    1C1148    2R1038             -1.   R1039               1.
    C1149     R0000             .183   R1040               1.
    C1149     R1046               1.   R1075             -.28
    C1149     R1076            -.463   R1077       -64.799999
    C1149     R1078             24.6
    C1150     R0000             .091   R1040               1.
    C1150     R1050               1.   R1079           -17.79
    C1150     R1080            -.618   R1081       -64.799999
    C1150     R1082             24.6
    C1151     R0000             .156   R1041               1.
    C1151     R1046               1.   R1075            -3.76
    C1151     R1076            -.397   R1077             68.5
    C1151     R1078            -51.3
    C1152     R0000             .078   R1041               1.
    C1152     R1050               1.   R1079           -18.69
    C1152     R1080            -.529   R1081             68.5
    C1152     R1082            -51.3
    C1154     R0000             .166   R1043               1.
    C1154     R1046               1.   R1075            -4.38
    C1154     R1076             -.42   R1077            21.25
    C1154     R1078            -24.4
    C1156     R0000             .344   R1035              -.2
    C1156     R1040               .2   R1043               .6
    C1156     R1045               1.
    C1158     R1046              -1.   R1047               1.
    C1158     R1075             1.32
    C1162     R1050              -1.   R1051               1.
    C1162     R1079              6.9
    C1163     R1018            -.196   R1020            -.285
    C1163     1R1021           -.377   R1023             -.06
    C1163     R1024            -.082   R1061               1.
    1C1163    R1018            -.468   R1020            -.241
    1C1163    1R1021           -.163   R1023            -.093
    1C1163    R1024            -.035   R1061               1.
    2C1163    R1018            -.438   1R1019           -.197
    2C1163    1R1021           -.365   R1061               1.
    3C1163    R1018             -.31   1R1020            -.31
    3C1163    2R1021            -.38   R1061               1.
    C1165     R1062              -1.   2R1062              1.
    C1165     R1091            -1.47   R1092              2.5
    C1165     R1093            -13.8   R1095              -5.
    C1166     R1063              -1.   1R1063              1.
    C1166     R1097            -3.56   R1098               3.
    C1166     R1101            -7.36
    1C1166    R1063              -1.   2R1063              1.
    1C1166    R1097            -2.83   R1098               3.
    1C1166    R1099             -8.5   R1101            -7.36
    C1167     R0000             .591   R1075            -4.46
    C1167     R1076               1.
    C1168     R0000             .591   R1079            -11.2
    C1168     R1080               1.
    C1F28     R1064              -1.   1R1064              1.
    C1F28     R1103               2.
    C1F35     R1068              -1.   R1069               1.
    C1F35     R1111             1.83   R1112            -1.98
    C1F35     R1113              5.5   R1115               3.
    C1F64     R1068              -1.   1R1068              1.
    C1F64     R1112              .15   R1113              5.5
    C1F64     R1115               3.
    CJ001     1RJ006            1.01   RJ007             .019
    CJ001     1RJ007            .006   1RJ008             .06
    CJ001     RJ010             .083   RJ012             .066
    CJ001     RJ013             .141   RJ014             .016
    CJ001     RJ018             .192   RJ019             .028
    CJ001     RJ024             .389   RJ031           -.0234
    CJ001     RJ048            1.222
    CJ002     RJ001             1.01   RJ007             .014
    CJ002     1RJ007            .006   1RJ008            .049
    CJ002     RJ010             .081   RJ012             .044
    CJ002     RJ013             .091   RJ014              .01
    CJ002     RJ015             .136   RJ016             .024
    CJ002     RJ025             .545   RJ031           -.0202
    CJ002     RJ048               1.
    2CJ002    1RJ001            1.01   RJ007             .013
    2CJ002    1RJ007            .005   1RJ008            .074
    2CJ002    RJ010             .143   RJ012             .039
    2CJ002    RJ013             .082   RJ014             .009
    2CJ002    RJ015             .122   RJ016             .022
    2CJ002    RJ025             .491   RJ031           -.0226
    2CJ002    RJ048             1.21
    CJ003     RJ002             1.01   RJ007             .013
    CJ003     1RJ007            .005   RJ008             .047
    CJ003     RJ009             .107   1RJ011             .05
    CJ003     RJ014             .124   RJ018             .164
    CJ003     RJ019             .027   RJ026             .463
    CJ003     RJ031            -.022   RJ048            1.052
    CJ004     RJ003             1.01   RJ007             .017
    CJ004     1RJ007            .007   1RJ008            .059
    CJ004     RJ010             .098   RJ012             .051
    CJ004     RJ013             .105   RJ014             .012
    CJ004     RJ015             .149   RJ016             .024
    CJ004     RJ027             .478   RJ031           -.0215
    CJ004     RJ048            1.043
    CJ005     RJ004             1.01   RJ007             .024
    CJ005     1RJ007            .009   1RJ008            .078
    CJ005     RJ010             .127   RJ012             .065
    CJ005     RJ013              .13   RJ014             .015
    CJ005     RJ020             .172   RJ021             .025
    CJ005     RJ028             .355   RJ031           -.0234
    CJ005     RJ048            1.235
    CJ006     RJ005             1.01   RJ007             .015
    CJ006     1RJ007            .003   RJ008             .047
    CJ006     RJ009             .097   RJ011             .049
    CJ006     RJ014             .107   RJ018             .149
    CJ006     RJ019             .023   RJ029              .51
    CJ006     RJ031           -.0203   RJ048               1.
    1CJ006    RJ005             1.01   RJ007             .015
    1CJ006    1RJ007            .003   RJ008             .047
    1CJ006    RJ009             .097   1RJ011            .049
    1CJ006    RJ014             .107   RJ018             .149
    1CJ006    RJ019             .023   RJ029              .51
    1CJ006    RJ031            -.023   RJ048               1.
    1CJ006    RJ100            .0464
    CJ007     RJ006             1.01   RJ007             .012
    CJ007     1RJ007            .004   RJ008              .06
    CJ007     RJ010             .103   RJ012             .061
    CJ007     RJ013             .126   RJ014             .014
    CJ007     RJ020             .172   RJ021              .03
    CJ007     RJ030             .418   RJ031            -.022
    CJ007     RJ048            1.142
    CJ008     RJ007              -1.   RJ031             1.16
    1CJ008    1RJ007             -1.   RJ031              1.2
    CJ009     R0000             .251   RJ007              -1.
    CJ009     RJ032               1.   RJ052            -7.41
    CJ009     RJ053            -96.5   RJ054         214.1001
    CJ009     RJ095            -.541
    CJ010     R0000             .126   RJ007              -1.
    CJ010     RJ033               1.   RJ055           -26.03
    CJ010     RJ056           -96.54   RJ057         214.1001
    CJ010     RJ096            -.734
    CJ012     RJ008               1.   1RJ008             -1.
    CJ013     R0000             .219   RJ008              -1.
    CJ013     RJ032               1.   1RJ050              1.
    CJ013     RJ052             34.5   RJ053         -76.6001
    CJ013     RJ054            33.02   RJ095            -.471
    CJ014     R0000             .109   RJ008              -1.
    CJ014     RJ033               1.   1RJ050              1.
    CJ014     RJ055            22.52   RJ056       -76.549999
    CJ014     RJ057            33.02   RJ096            -.639
    CJ016     R0000               .1   RJ010              -1.
    CJ016     RJ033               1.   1RJ050              1.
    CJ016     RJ055            31.26   RJ056            30.29
    CJ016     RJ057           -42.14   RJ096            -.586
    1CJ016    R0000             .091   RJ009              -1.
    1CJ016    RJ033               1.   1RJ050              1.
    1CJ016    RJ055            31.26   RJ056            30.29
    1CJ016    RJ057           -42.14   RJ096            -.586
    CJ018     RJ007             -.04   RJ008             -.96
    CJ018     RJ035               1.
    1CJ018    RJ007            -.014   RJ008            -.336
    1CJ018    RJ009             -.65   RJ035               1.
    CJ019     RJ007            -.014   RJ008            -.336
    CJ019     RJ010             -.65   RJ035               1.
    CJ020     R0000           1.2666   RC0ST            2.004
    CJ020     RLDEX              -1.   RJ007            -.014
    CJ020     RJ008            -.336   RJ010             -.65
    CJ020     RJ036               1.
    1CJ020    R0000           1.2666   RC0ST            2.004
    1CJ020    RLDEX              -1.   RJ007             -.02
    1CJ020    RJ008             -.48   RJ009              -.5
    1CJ020    RJ036               1.   1RJ050              .5
    CJ021     R0000           1.2666   RC0ST            2.004
    CJ021     RLDEX              -1.   RJ007             -.02
    CJ021     RJ008             -.48   RJ010              -.5
    CJ021     RJ036               1.
    1CJ021    RJ008              -.3   RJ010              -.7
    1CJ021    1RJ036              1.
    2CJ021    RJ010              -.7   RJ011              -.3
    2CJ021    1RJ036              1.
    3CJ021    RJ008              -.3   RJ009              -.7
    3CJ021    1RJ036              1.
    4CJ021    RJ009              -.7   RJ011              -.3
    4CJ021    1RJ036              1.
    CJ022     RJ031             1.11   RJ035              -1.
    CJ023     R0000            .0931   RJ007             .101
    CJ023     1RJ007            .152   RJ008              .01
    CJ023     RJ009              -1.   RJ011              .04
    CJ023     RJ031            -.101   RJ049            1.157
    CJ023     1RJ050              1.   RJ097             .697
    CJ023     RJ099             -.19
    CJ025     R0000            .0829   RJ007             .074
    CJ025     1RJ007            .112   RJ008              .01
    CJ025     RJ009              -1.   RJ011              .04
    CJ025     RJ031           -.0956   RJ049             .909
    CJ025     1RJ050              1.   1RJ097            .764
    CJ025     RJ099             -.19
    CJ028     R0000             .096   RJ033               1.
    CJ028     RJ055             -14.   RJ056            -9.25
    CJ028     RJ057            -9.25   RJ096            -.562
    CJ028     1RJ097             -1.
    CJ029     R0000             .193   RJ032               1.
    CJ029     RJ052             -.66   RJ053            -9.25
    CJ029     RJ054            -9.25   RJ095            -.414
    CJ029     1RJ097             -1.
    CJ030     R0000            .1209   RJ007             .099
    CJ030     1RJ007            .148   RJ008              .01
    CJ030     RJ010              -1.   RJ011              .04
    CJ030     RJ031           -.0956   RJ049               1.
    CJ030     1RJ097            .703   RJ099             -.19
    CJ031     R0000            .1289   RJ007             .126
    CJ031     1RJ007            .167   RJ008              .01
    CJ031     RJ010              -1.   RJ011              .04
    CJ031     RJ031            -.101   RJ049              1.3
    CJ031     RJ051              .02   RJ097             .637
    CJ031     RJ099             -.19
    CJ032     R0000            .1084   RJ007             .126
    CJ032     1RJ007            .187   RJ008              .01
    CJ032     RJ010              -1.   RJ011              .04
    CJ032     RJ031            -.101   RJ049            1.445
    CJ032     RJ097             .637   RJ099             -.19
    CJ033     R0000              .19   RJ032               1.
    CJ033     RJ052            -5.34   RJ053            -5.21
    CJ033     RJ054            -4.04   RJ095            -.408
    CJ033     RJ097              -1.
    CJ034     R0000             .096   RJ033               1.
    CJ034     RJ055           -13.49   RJ056             3.97
    CJ034     RJ057           -14.42   RJ096            -.563
    CJ034     1RJ098             -1.
    CJ035     R0000            .0707   RJ007             .064
    CJ035     1RJ007            .094   RJ008              .01
    CJ035     RJ009            -.667   1RJ011           -.333
    CJ035     RJ014              .04   RJ031           -.0956
    CJ035     RJ049             .808   1RJ050            .667
    CJ035     1RJ098             .79   RJ099             -.19
    CJ036     R0000            .0804   RJ007             .089
    CJ036     1RJ007            .135   RJ008              .01
    CJ036     RJ009            -.667   1RJ011           -.333
    CJ036     RJ014              .04   RJ031            -.101
    CJ036     RJ049            1.044   1RJ050            .667
    CJ036     RJ098             .726   RJ099             -.19
    CJ037     RJ011               1.   RJ012              -1.
    1CJ037    RJ011               1.   1RJ011             -1.
    CJ039     RJ011              -1.   RJ040               1.
    CJ039     RJ061               8.   RJ062             -8.5
    CJ039     RJ063            36.08   RJ064             -20.
    CJ039     RJ065             -13.   RJ079               1.
    CJ040     RJ011              -1.   RJ043               1.
    CJ040     RJ066             10.2   RJ067            -13.5
    CJ040     RJ068            36.08   RJ069             -30.
    CJ040     RJ070            13.12   RJ071             -26.
    CJ040     RJ079               1.
    CJ041     RJ011              -1.   RJ045               1.
    CJ041     RJ072             -38.   RJ073           -42.73
    CJ041     1RJ073           41.92   RJ074            36.08
    CJ041     1RJ074            -95.
    CJ042     RJ011              -1.   RJ046               1.
    CJ042     RJ075             -38.   RJ076           -39.61
    CJ042     1RJ076           38.97   RJ077            36.08
    CJ042     RJ078             -92.
    1CJ042    RJ011              -1.   RJ042               1.
    1CJ042    RJ102             -23.   RJ103           -39.61
    1CJ042    RJ104            38.97   RJ105            36.08
    1CJ042    RJ106             -92.
    CJ043     R0000             .191   RJ032               1.
    CJ043     RJ052            -4.84   RJ053             2.61
    CJ043     RJ054           -10.19   RJ095             -.41
    CJ043     RJ098              -1.
    CJ044     R0000             .193   RJ032               1.
    CJ044     RJ052              .13   RJ053             3.97
    CJ044     RJ054           -14.42   RJ095            -.415
    CJ044     1RJ098             -1.
    CJ045     R0000            .0933   RJ007             .076
    CJ045     1RJ007            .113   RJ008              .01
    CJ045     RJ010            -.667   RJ012            -.333
    CJ045     RJ014              .04   RJ031           -.0956
    CJ045     RJ049              .88   1RJ098            .761
    CJ045     RJ099             -.19
    CJ046     R0000            .1074   RJ007             .103
    CJ046     1RJ007            .144   RJ008              .01
    CJ046     RJ010            -.667   RJ012            -.333
    CJ046     RJ014              .04   RJ031            -.101
    CJ046     RJ049            1.111   RJ051             .009
    CJ046     RJ098             .694   RJ099             -.19
    1CJ046    R0000            .0949   RJ007             .103
    1CJ046    1RJ007            .153   RJ008              .01
    1CJ046    RJ010            -.667   RJ012            -.333
    1CJ046    RJ014              .04   RJ031           -.0956
    1CJ046    RJ049            1.182   RJ098             .694
    1CJ046    RJ099             -.19
    CJ047     RJ012             -.44   RJ013             -.56
    CJ047     RJ034               1.   RJ038               1.
    1CJ047    1RJ011            -.44   RJ013             -.56
    1CJ047    RJ034               1.   RJ038               1.
    1CJ048    RJ011               .1   RJ013             -1.6
    1CJ048    RJ015               .5   RJ034               1.
    1CJ048    1RJ037              1.   2RJ037             -1.
    CJ049     RJ013              -1.   RJ014               1.
    CJ050     RJ011             -.15   RJ014             -.85
    CJ050     RJ031            -.035   RJ034              .81
    CJ050     RJ039              .81   RJ040              .19
    CJ050     RJ050               1.   RJ061            -1.48
    CJ050     RJ062              .48   RJ063              .72
    CJ050     RJ064             -3.8   RJ065              9.7
    CJ050     RJ079              .19
    CJ051     RJ011             -.15   RJ014             -.85
    CJ051     RJ031            -.035   RJ034              .81
    CJ051     RJ039              .81   RJ043              .19
    CJ051     RJ050               1.   RJ066            -1.06
    CJ051     RJ067             -.48   RJ068              .74
    CJ051     RJ069             -5.7   RJ070             1.97
    CJ051     RJ071              7.5   RJ079              .19
    1CJ051    RJ011             -.68   RJ014              -1.
    1CJ051    RJ031            -.035   RJ034             1.55
    1CJ051    RJ038             1.55   RJ043              .13
    1CJ051    RJ050               1.   RJ066             -.96
    1CJ051    RJ067              .46   RJ068             1.71
    1CJ051    RJ069             -3.9   RJ070             1.21
    1CJ051    RJ071             4.94   RJ079              .13
    CJ052     RJ011             -.15   RJ014             -.85
    CJ052     RJ031            -.035   RJ034              .81
    CJ052     RJ039              .81   RJ045              .19
    CJ052     RJ050               1.   RJ072            -5.13
    CJ052     RJ073            -6.12   1RJ073            5.96
    CJ052     RJ074              .74   1RJ074           -18.1
    1CJ052    RJ011             -.15   RJ014             -.85
    1CJ052    RJ031            -.035   RJ034              .81
    1CJ052    RJ039              .81   RJ042              .19
    1CJ052    RJ050               1.   RJ102            -2.28
    1CJ052    RJ103            -5.53   RJ104              5.4
    1CJ052    RJ105              .74   RJ106            -17.5
    CJ053     RJ011             -.15   RJ014             -.85
    CJ053     RJ031            -.035   RJ034              .81
    CJ053     RJ039              .81   RJ046              .19
    CJ053     RJ050               1.   RJ075            -5.13
    CJ053     RJ076            -5.53   1RJ076             5.4
    CJ053     RJ077              .74   RJ078            -17.5
    1CJ053    RJ011             -.68   RJ014              -1.
    1CJ053    RJ031            -.035   RJ034             1.55
    1CJ053    RJ038             1.55   RJ046              .13
    1CJ053    RJ050               1.   RJ075            -2.34
    1CJ053    RJ076            -3.51   1RJ076            3.42
    1CJ053    RJ077              .08   RJ078             -12.
    2CJ053    RJ011             -.68   RJ014              -1.
    2CJ053    RJ031            -.035   RJ034             1.55
    2CJ053    RJ038             1.55   RJ042              .13
    2CJ053    RJ050               1.   RJ102             -.39
    2CJ053    RJ103            -3.51   RJ104             3.42
    2CJ053    RJ105              .08   RJ106             -12.
    CJ054     RJ014              -1.   RJ040               1.
    CJ054     RJ061              3.8   RJ062             -6.5
    CJ054     RJ063              .71   RJ064             -13.
    CJ054     RJ065             -13.   RJ079               1.
    CJ055     RJ014              -1.   RJ043               1.
    CJ055     RJ066               6.   RJ067            -11.5
    CJ055     RJ068              .71   RJ069             -23.
    CJ055     RJ070             8.24   RJ071             -26.
    CJ055     RJ079               1.
    CJ056     RJ014              -1.   RJ045               1.
    CJ056     RJ072             -36.   RJ073           -30.09
    CJ056     1RJ073           29.28   RJ074              .71
    CJ056     1RJ074           -76.5
    1CJ056    RJ014              -1.   RJ031            1.087
    CJ057     RJ014              -1.   RJ046               1.
    CJ057     RJ075             -36.   RJ076           -26.97
    CJ057     1RJ076           26.33   RJ077              .71
    CJ057     RJ078            -73.5
    1CJ057    RJ014              -1.   RJ042               1.
    1CJ057    RJ102             -21.   RJ103           -26.97
    1CJ057    RJ104            26.33   RJ105              .71
    1CJ057    RJ106            -73.5
    CJ058     RJ015              -1.   RJ040               1.
    CJ058     RJ061             -2.6   RJ062               3.
    CJ058     RJ063            -3.01   RJ064               4.
    CJ058     RJ065              -8.   RJ079               1.
    1CJ058    RJ015              -1.   RJ031            -.055
    1CJ058    RJ040               1.   RJ061             -2.6
    1CJ058    RJ062              -.5   RJ063            -3.01
    1CJ058    RJ064               4.   RJ065              -8.
    1CJ058    RJ079               1.   RJ100               1.
    CJ059     RJ015              -1.   RJ043               1.
    CJ059     RJ066              -.4   RJ067              -2.
    CJ059     RJ068            -3.33   RJ069              -6.
    CJ059     RJ070              .38   RJ071             -21.
    CJ059     RJ079               1.
    CJ060     RJ015              -1.   RJ045               1.
    CJ060     RJ072             -27.   RJ073           -22.23
    CJ060     1RJ073           21.42   RJ074            -3.33
    CJ060     1RJ074            -53.
    1CJ060    RJ015              -1.   RJ031             1.06
    CJ061     RJ015              -1.   RJ046               1.
    CJ061     RJ075             -27.   RJ076           -19.11
    CJ061     1RJ076           18.47   RJ077            -3.33
    CJ061     RJ078             -50.
    1CJ061    RJ015              -1.   RJ042               1.
    1CJ061    RJ102             -12.   RJ103           -19.11
    1CJ061    RJ104            18.47   RJ105            -3.33
    1CJ061    RJ106             -50.
    CJ062     RJ018              -1.   RJ040               1.
    CJ062     RJ061             -3.8   RJ062               .1
    CJ062     RJ063            -3.01   RJ064               7.
    CJ062     RJ065             -6.5   RJ079               1.
    CJ063     RJ018              -1.   RJ043               1.
    CJ063     RJ066             -1.6   RJ067             -4.9
    CJ063     RJ068            -3.33   RJ069              -3.
    CJ063     RJ070            -1.22   RJ071            -18.5
    CJ063     RJ079               1.
    CJ064     RJ018              -1.   RJ045               1.
    CJ064     RJ072            -29.4   RJ073           -20.63
    CJ064     1RJ073           19.82   RJ074            -3.33
    CJ064     1RJ074            -50.
    CJ065     RJ018              -1.   RJ046               1.
    CJ065     RJ075            -29.4   RJ076           -17.51
    CJ065     1RJ076           16.87   RJ077            -3.33
    CJ065     RJ078             -47.
    1CJ065    RJ018              -1.   RJ042               1.
    1CJ065    RJ102            -14.4   RJ103           -17.51
    1CJ065    RJ104            16.87   RJ105            -3.33
    1CJ065    RJ106             -47.
    CJ066     RJ020              -1.   RJ040               1.
    CJ066     RJ061             -2.7   RJ062               .3
    CJ066     RJ063            -3.01   RJ064               4.
    CJ066     RJ065              -4.   RJ079               1.
    CJ067     RJ020              -1.   RJ043               1.
    CJ067     RJ066              -.5   RJ067             -4.7
    CJ067     RJ068            -3.17   RJ069              -6.
    CJ067     RJ070             -.92   RJ071             -20.
    CJ067     RJ079               1.
    CJ068     RJ020              -1.   RJ045               1.
    CJ068     RJ072            -29.2   RJ073           -20.93
    CJ068     1RJ073           20.12   RJ074            -3.17
    CJ068     1RJ074            -53.
    CJ069     RJ020              -1.   RJ046               1.
    CJ069     RJ075            -29.2   RJ076           -17.81
    CJ069     1RJ076           17.17   RJ077            -3.17
    CJ069     RJ078             -50.
    CJ070     RJ020              -1.   RJ042               1.
    CJ070     RJ102            -14.2   RJ103           -17.81
    CJ070     RJ104            17.17   RJ105            -3.17
    CJ070     RJ106             -50.
    CJ074     RJ016              -1.   RJ040               1.
    CJ074     RJ061             -7.6   RJ062              11.
    CJ074     RJ063            -1.65   RJ064              26.
    CJ074     RJ065              -4.   RJ079               1.
    CJ075     RJ016              -1.   RJ043               1.
    CJ075     RJ066             -5.4   RJ067               6.
    CJ075     RJ068            -1.65   RJ069              16.
    CJ075     RJ070            -7.62   RJ071             -14.
    CJ075     RJ079               1.
    CJ076     RJ016              -1.   RJ045               1.
    CJ076     RJ072            -18.5   RJ073           -14.23
    CJ076     1RJ073           13.42   RJ074            -1.65
    CJ076     1RJ074            -30.
    1CJ076    RJ016              -1.   RJ031            1.044
    CJ077     RJ016              -1.   RJ046               1.
    CJ077     RJ075            -18.5   RJ076           -11.11
    CJ077     1RJ076           10.47   RJ077            -1.65
    CJ077     RJ078             -27.
    1CJ077    RJ016              -1.   RJ042               1.
    1CJ077    RJ102             -3.5   RJ103           -11.11
    1CJ077    RJ104            10.47   RJ105            -1.65
    1CJ077    RJ106             -27.
    CJ078     RJ019              -1.   RJ040               1.
    CJ078     RJ061             -8.7   RJ062              5.5
    CJ078     RJ063            -1.65   RJ064              38.
    CJ078     RJ065              -2.   RJ079               1.
    CJ079     RJ019              -1.   RJ043               1.
    CJ079     RJ066             -6.5   RJ067               .5
    CJ079     RJ068            -1.65   RJ069              28.
    CJ079     RJ070            -9.12   RJ071            -11.5
    CJ079     RJ079               1.
    CJ080     RJ019              -1.   RJ045               1.
    CJ080     RJ072             -24.   RJ073           -12.73
    CJ080     1RJ073           11.92   RJ074            -1.65
    CJ080     1RJ074            -18.
    CJ081     RJ019              -1.   RJ046               1.
    CJ081     RJ075             -24.   RJ076            -9.61
    CJ081     1RJ076            8.97   RJ077            -1.65
    CJ081     RJ078             -15.
    CJ082     RJ019              -1.   RJ042               1.
    CJ082     RJ102              -9.   RJ103            -9.61
    CJ082     RJ104             8.97   RJ105            -1.65
    CJ082     RJ106             -15.
    CJ086     RJ021              -1.   RJ040               1.
    CJ086     RJ061             -6.3   RJ062               8.
    CJ086     RJ063            -1.65   RJ064              38.
    CJ086     RJ065               3.   RJ079               1.
    CJ087     RJ021              -1.   RJ043               1.
    CJ087     RJ066             -5.1   RJ067               3.
    CJ087     RJ068            -1.65   RJ069              28.
    CJ087     RJ070            -6.52   RJ071            -12.5
    CJ087     RJ079               1.
    CJ088     RJ021              -1.   RJ045               1.
    CJ088     RJ072            -21.5   RJ073           -15.33
    CJ088     1RJ073           14.52   RJ074            -1.65
    CJ088     1RJ074            -18.
    CJ089     RJ021              -1.   RJ046               1.
    CJ089     RJ075            -21.5   RJ076           -12.21
    CJ089     1RJ076           11.57   RJ077            -1.65
    CJ089     RJ078             -15.
    CJ090     RJ021              -1.   RJ042               1.
    CJ090     RJ102             -6.5   RJ103           -12.21
    CJ090     RJ104            11.57   RJ105            -1.65
    CJ090     RJ106             -15.
    CJ099     RJ025              -1.   RJ045               1.
    CJ099     RJ072              1.5   RJ073              .21
    CJ099     1RJ073           -1.02   RJ074            -2.14
    CJ099     1RJ074            -10.
    CJ100     RJ025              -1.   RJ046               1.
    CJ100     RJ075              1.5   RJ076             3.33
    CJ100     1RJ076           -3.97   RJ077            -2.14
    CJ100     RJ078              -7.
    1CJ100    RJ025              -1.   RJ042               1.
    1CJ100    RJ102             16.5   RJ103             3.33
    1CJ100    RJ104            -3.97   RJ105            -2.14
    1CJ100    RJ106              -7.
    CJ102     RJ026              -1.   RJ045               1.
    CJ102     RJ072             -15.   RJ073             -1.8
    CJ102     1RJ073             .99   RJ074            -2.14
    CJ102     1RJ074             -2.
    CJ103     RJ026              -1.   RJ046               1.
    CJ103     RJ075             -15.   RJ076             1.32
    CJ103     1RJ076           -1.96   RJ077            -2.14
    CJ103     RJ078               1.
    1CJ103    RJ026              -1.   RJ042               1.
    1CJ103    RJ103             1.32   RJ104            -1.96
    1CJ103    RJ105            -2.14   RJ106               1.
    CJ105     RJ027              -1.   RJ045               1.
    CJ105     RJ072             -3.5   RJ073            -1.65
    CJ105     1RJ073             .84   RJ074            -2.14
    CJ105     1RJ074            -10.
    CJ106     RJ027              -1.   RJ046               1.
    CJ106     RJ075             -3.5   RJ076             1.47
    CJ106     1RJ076           -2.11   RJ077            -2.14
    CJ106     RJ078              -7.
    CJ107     RJ027              -1.   RJ042               1.
    CJ107     RJ102             11.5   RJ103             1.47
    CJ107     RJ104            -2.11   RJ105            -2.14
    CJ107     RJ106              -7.
    CJ108     RJ028              -1.   RJ045               1.
    CJ108     RJ072             -14.   RJ073            -4.13
    CJ108     1RJ073            3.32   RJ074            -2.14
    CJ108     1RJ074             10.
    CJ109     RJ028              -1.   RJ046               1.
    CJ109     RJ075             -14.   RJ076            -1.01
    CJ109     1RJ076             .37   RJ077            -2.14
    CJ109     RJ078              13.
    1CJ109    RJ028              -1.   RJ042               1.
    1CJ109    RJ102               1.   RJ103            -1.01
    1CJ109    RJ104              .37   RJ105            -2.14
    1CJ109    RJ106              13.
    CJ111     RJ029              -1.   RJ045               1.
    CJ111     RJ072             -11.   RJ073               .8
    CJ111     1RJ073           -1.61   RJ074            -2.14
    CJ111     1RJ074             -2.
    CJ112     RJ029              -1.   RJ046               1.
    CJ112     RJ075             -11.   RJ076             3.92
    CJ112     1RJ076           -4.56   RJ077            -2.14
    CJ112     RJ078               1.
    CJ113     RJ029              -1.   RJ042               1.
    CJ113     RJ102               4.   RJ103             3.92
    CJ113     RJ104            -4.56   RJ105            -2.14
    CJ113     RJ106               1.
    CJ114     RJ030              -1.   RJ045               1.
    CJ114     RJ072             -11.   RJ073            -3.72
    CJ114     1RJ073            2.91   RJ074            -2.14
    CJ114     1RJ074             -2.
    CJ115     RJ030              -1.   RJ046               1.
    CJ115     RJ075             -11.   RJ076              -.6
    CJ115     1RJ076            -.04   RJ077            -2.14
    CJ115     RJ078               1.
    1CJ115    RJ030              -1.   RJ042               1.
    1CJ115    RJ102               4.   RJ103              -.6
    1CJ115    RJ104             -.04   RJ105            -2.14
    1CJ115    RJ106               1.
    CJ116     RJ024              -1.   RJ045               1.
    CJ116     RJ072             -22.   RJ073            -5.87
    CJ116     1RJ073            5.06   RJ074            -2.14
    CJ116     1RJ074             17.
    CJ117     RJ024              -1.   RJ046               1.
    CJ117     RJ075             -22.   RJ076            -2.75
    CJ117     1RJ076            2.11   RJ077            -2.14
    CJ117     RJ078              20.
    CJ118     RJ024              -1.   RJ042               1.
    CJ118     RJ102              -7.   RJ103            -2.75
    CJ118     RJ104             2.11   RJ105            -2.14
    CJ118     RJ106              20.
    1CJ120    RJ014             -.05   RJ015             -.33
    1CJ120    RJ025             -.62   RJ047               1.
    CJ121     1RJ007             -1.   RJ051               1.
    CJ122     RJ031               1.   RJ045              -1.
    1CJ122    RJ031               1.   RJ046              -1.
    CJ123     R0000            .9103   RC0ST            1.258
    CJ123     RJ033              -1.   RJ055            2.667
    CJ123     RJ081               1.
    CJ124     RJ032             -.65   RJ033             -.35
    CJ124     RJ053              2.6   RJ054            -6.07
    CJ124     RJ056              1.4   RJ057            -3.27
    CJ124     RJ082               1.
    1CJ125    R0000             .165   RJ032              -1.
    1CJ125    RJ052             5.73   RJ053               4.
    1CJ125    RJ054            -9.33   RJ083               1.
    CJ126     R0000           1.0891   RC0ST            1.258
    CJ126     RJ038              -1.   RJ084               1.
    1CJ126    R0000            1.079   RC0ST            1.258
    1CJ126    RJ040              -1.   1RJ084              1.
    CJ131     RJ046              -1.   RJ075               6.
    CJ131     RJ076             1.23   1RJ076           -1.77
    CJ131     RJ078              -3.   RJ089               1.
    CJ133     R0000             .001   RJ048               1.
    CJ133     RJ080              -1.
    CJ134     RJ045              -1.   RJ073              .74
    CJ134     1RJ073            -.78   RJ091               1.
    CJ135     RJ045              -1.   RJ072               5.
    CJ135     RJ073             1.96   1RJ073           -1.84
    CJ135     1RJ074             11.   RJ092               1.
    CJ150     R0000             .651   RJ052           -8.035
    CJ150     RJ095               1.
    CJ151     R0000             .651   RJ055          -15.625
    CJ151     RJ096               1.
    CJG55     RJ040              -1.   1RJ041              1.
    CJG55     RJ062              2.5
    CJF35     RJ046              -1.   2RJ046              1.
    CJF35     RJ075               5.
    CJF80     RJ045              -1.   2RJ045              1.
    CJF80     RJ072               5.   1RJ073             .26
    CJF80     1RJ074              3.
    CM001     R0000            .8675   1RM001              1.
    CM001     RM002             .155   2RM003             .24
    CM001     1RM006            .555   RGACH               .5
    CM001     RKUWT               .5
    1CM004    2RM003           -.145   RM005               1.
    1CM004    1RM006           -.855
    CM005     R0000            .8687   1RM001              1.
    CM005     RM002              .15   2RM003            .255
    CM005     1RM006            .545   RKUWT               1.
    CM006     1RM003              1.   2RM003          -1.171
    CM006     RM004             .171
    CM007     R0000            .8677   1RM001              1.
    CM007     RM002             .155   2RM003             .24
    CM007     RM006             .416   2RM006            .139
    CM007     RGACH               .5   RKUWT               .5
    CY001     R0000             .275   RY001            -.149
    CY001     RY002            -.273   RY004             -.55
    CY001     RY005               1.   RKUWT               1.
    1CY001    R0000             .275   RY001            -.173
    1CY001    RY002            -.245   RY004             -.55
    1CY001    RY005               1.   RKUWT               1.
    CY002     R0000             -2.4   RY001               1.
    CY003     R0000             -2.4   1RY001              1.
    CZ001     RZ001              -1.   RZ002              1.4
    CZ001     RZ004              -.3   RZ005               .2
    CZ001     RZ006             -.25   RZ019             -.05
    CZ013     RZ002              .16   RZ004              -.1
    CZ013     RZ005               .9   RZ006              -1.
    CZ013     RZ016              .03   RZ019              .01
    CZ014     RZ016               .2   RZ019               .8
    CZ014     RZ020              -1.
    ZH145     R0000            1.798   RC0ST           2.5667
    ZH145     RMSEX              -1.   RH031               1.
    ZH145     RH106               1.   RZ001               1.
    HM145     R0000           1.0821   RC0ST            2.133
    HM145     RMSEX               1.   RH031              -1.
    HM145     RM011               1.
    ZM145     R0000           2.3051   RC0ST            3.642
    ZM145     RM011               1.   RZ001               1.
    ZH130     R0000            1.665   RC0ST           2.5667
    ZH130     RMSEX              -1.   RH032               1.
    ZH130     RH106               1.   RZ002               1.
    HM130     R0000           1.0821   RC0ST            2.133
    HM130     RMSEX               1.   RH032              -1.
    HM130     RM012               1.
    ZM130     R0000           2.2078   RC0ST            3.483
    ZM130     RM012               1.   RZ002               1.
    HBBMS     R0000            .3197   RC0ST             .629
    HBBMS     RB033               1.   1RH034             -1.
    BDBMS     R0000            .2691   RC0ST              .44
    BDBMS     RB033              -1.   RDBMS               1.
    HDBMS     R0000            .1698   RC0ST             .335
    HDBMS     RDBMS               1.   1RH034             -1.
    H1BMS     R0000            .3102   RC0ST             .492
    H1BMS     1RH034             -1.   R1045               1.
    HBPMS     R0000            .3196   RC0ST            .6292
    HBPMS     RB035               1.   1RH035             -1.
    JBPMS     R0000            1.301   RC0ST            2.038
    JBPMS     RMSEX              -1.   RB035               1.
    JBPMS     RJ083              -1.
    BHPMS     R0000            .2875   RC0ST               .5
    BHPMS     RB035              -1.   1RH035              1.
    BHPMS     RH106               1.
    1HPMS     R0000             .256   RC0ST             .377
    1HPMS     1RH035              1.   RH106               1.
    1HPMS     R1047              -1.
    JHPMS     R0000           1.2071   RC0ST            1.883
    JHPMS     RMSEX              -1.   1RH035              1.
    JHPMS     RH106               1.   RJ083              -1.
    H1PMS     R0000            .3102   RC0ST             .492
    H1PMS     1RH035             -1.   R1047               1.
    J1PMS     R0000           1.4161   RC0ST            2.112
    J1PMS     RMSEX              -1.   R1047               1.
    J1PMS     RJ083              -1.
    HBRMS     R0000            .3196   RC0ST            .6292
    HBRMS     RB037               1.   1RH036             -1.
    JBRMS     R0000            1.301   RC0ST            2.038
    JBRMS     RMSEX              -1.   RB037               1.
    JBRMS     RJ082              -1.
    BHRMS     R0000            .2875   RC0ST               .5
    BHRMS     RB037              -1.   1RH036              1.
    BHRMS     RH106               1.
    1HRMS     R0000             .256   RC0ST             .377
    1HRMS     1RH036              1.   RH106               1.
    1HRMS     R1051              -1.
    JHRMS     R0000           1.2071   RC0ST            1.883
    JHRMS     RMSEX              -1.   1RH036              1.
    JHRMS     RH106               1.   RJ082              -1.
    H1RMS     R0000            .3102   RC0ST             .492
    H1RMS     1RH036             -1.   R1051               1.
    J1RMS     R0000           1.4161   RC0ST            2.112
    J1RMS     RMSEX              -1.   R1051               1.
    J1RMS     RJ082              -1.
    HNM00     R0000            .2909   30M00               1.
    HNM00     RC0ST             .575   RMSEX               1.
    HNM00     RH034              -1.   RH063           -2.281
    HNM00     1RH066             .15   2RH106              1.
    ZBM83     R0000           1.8441   RC0ST            2.708
    ZBM83     RMSEX              -1.   RB036               1.
    ZBM83     RB068            4.107   RZ006               1.
    ZHM83     R0000            1.659   RC0ST            2.554
    ZHM83     RMSEX              -1.   RH036               1.
    ZHM83     RH072            4.138   RH106               1.
    ZHM83     RZ006               1.
    ZJM83     R0000            .6885   RC0ST            .9542
    ZJM83     RJ033               1.   RJ055             5.58
    ZJM83     RJ056             -8.3   RJ057              9.7
    ZJM83     RJ096            -.594   RZ006               1.
    EAM87     R0000           1.1876   KW1RC               1.
    EAM87     1RA099              1.   RE040              -1.
    AEM87     R0000           1.3354   KW1RC              -1.
    AEM87     1RA099             -1.   RE040               1.
    HNM90     R0000            .4099   30M91               1.
    HNM90     RC0ST             .575   RMSEX               1.
    HNM90     RH036              -1.   RH072            5.517
    HNM90     RH075             .154   2RH106              1.
    1NM90     R0000            .5832   30M91               1.
    1NM90     RC0ST             .729   RMSEX               1.
    1NM90     R1050              -1.   R1079              5.5
    1NM90     R1080             .154
    JNM90     R0000           1.6032   30M91               1.
    JNM90     RC0ST            2.429   RH106               1.
    JNM90     RJ032             -.58   RJ033             -.42
    JNM90     RJ053             2.32   RJ054            -5.41
    JNM90     RJ056             1.68   RJ057            -3.92
    1HNM94    R0000           1.0859   30M94               1.
    1HNM94    RC0ST             .575   RMSEX               1.
    1HNM94    RH008             -.02   RH013             -.04
    1HNM94    RH016             -.13   RH095             -.81
    1HNM94    2RH106              1.
    2HNM94    R0000           1.0859   30M94               1.
    2HNM94    RC0ST             .575   RMSEX               1.
    2HNM94    RH010             -.13   RH013             -.29
    2HNM94    RH095             -.58   2RH106              1.
    JNM94     R0000           1.6032   30M94               1.
    JNM94     RC0ST            2.429   RH106               1.
    JNM94     RJ032             -.92   RJ033             -.08
    JNM94     RJ053             3.68   RJ054            -8.59
    JNM94     RJ056              .32   RJ057             -.75
    ZHM95     R0000            1.659   RC0ST            2.554
    ZHM95     RMSEX              -1.   RH036               1.
    ZHM95     RH072          -12.195   RH106               1.
    ZHM95     RZ005               1.
    ZJM95     R0000            .7866   RC0ST             .954
    ZJM95     RJ032               1.   RJ052             5.48
    ZJM95     RJ053            -4.07   RJ054              9.5
    ZJM95     RJ095             -.43   RZ005               1.
    EAM97     R0000           1.1876   KW1RC               1.
    EAM97     RA097               1.   RE035              -1.
    EAM97     RE068            -1.31
    AEM97     R0000           1.3354   KW1RC              -1.
    AEM97     RA039              -1.   RE036               1.
    ABLDF     R0000           4.2007   RC0ST            2.133
    ABLDF     RLDEX              -1.   RA023              -1.
    ABLDF     RA100               1.   1RB038              1.
    EBLDF     R0000           4.3074   RC0ST            2.367
    EBLDF     RLDEX              -1.   1RB038              1.
    EBLDF     RE039              -1.   RE104               1.
    HBLDF     R0000            .3197   RC0ST             .629
    HBLDF     1RB038              1.   RH039              -1.
    1BLDF     R0000            .4124   RC0ST             .662
    1BLDF     1RB038              1.   R1055              -1.
    JBLDF     R0000            1.301   RC0ST            2.038
    JBLDF     RLDEX              -1.   1RB038              1.
    JBLDF     RJ035              -1.
    MBLDF     R0000            .6302   RC0ST            1.271
    MBLDF     RLDEX              -1.   1RB038              1.
    MBLDF     RM002              -1.
    YBLDF     R0000           1.5743   RC0ST            2.612
    YBLDF     RLDEX              -1.   1RB038              1.
    YBLDF     RY001               1.
    1YBLDF    R0000           1.5743   RC0ST            2.612
    1YBLDF    RLDEX              -1.   1RB038              1.
    1YBLDF    1RY001              1.
    BHLDF     R0000            .2875   RC0ST               .5
    BHLDF     1RB038             -1.   RH039               1.
    BHLDF     RH106               1.
    EHLDF     R0000           4.2174   RC0ST            2.221
    EHLDF     RLDEX              -1.   RE039              -1.
    EHLDF     RE104               1.   RH039               1.
    EHLDF     RH106               1.
    1HLDF     R0000             .256   RC0ST             .377
    1HLDF     RH039               1.   RH106               1.
    1HLDF     R1055              -1.
    JHLDF     R0000           1.2071   RC0ST            1.883
    JHLDF     RLDEX              -1.   RH039               1.
    JHLDF     RH106               1.   RJ035              -1.
    YHLDF     R0000           1.4803   RC0ST            2.458
    YHLDF     RLDEX              -1.   RH039               1.
    YHLDF     RH106               1.   RY001               1.
    1YHLDF    R0000           1.4803   RC0ST            2.458
    1YHLDF    RLDEX              -1.   RH039               1.
    1YHLDF    RH106               1.   1RY001              1.
    A1LDF     R0000           4.3906   RC0ST            2.367
    A1LDF     RLDEX              -1.   RA023              -1.
    A1LDF     RA100               1.   R1055               1.
    B1LDF     R0000            .4671   RC0ST             .717
    B1LDF     1RB038             -1.   R1055               1.
    E1LDF     R0000           4.4302   RC0ST            2.458
    E1LDF     RLDEX              -1.   RE039              -1.
    E1LDF     RE104               1.   R1055               1.
    H1LDF     R0000            .3102   RC0ST             .492
    H1LDF     RH039              -1.   R1055               1.
    J1LDF     R0000           1.4161   RC0ST            2.112
    J1LDF     RLDEX              -1.   R1055               1.
    J1LDF     RJ035              -1.
    M1LDF     R0000             .745   RC0ST            1.346
    M1LDF     RLDEX              -1.   R1055               1.
    M1LDF     RM002              -1.
    Y1LDF     R0000            1.689   RC0ST            2.688
    Y1LDF     RLDEX              -1.   R1055               1.
    Y1LDF     RY001               1.
    1Y1LDF    R0000            1.689   RC0ST            2.688
    1Y1LDF    RLDEX              -1.   R1055               1.
    1Y1LDF    1RY001              1.
    ACLD4     R0000           4.7156   RC0ST           2.4208
    ACLD4     RLDEX              -1.   RA023              -1.
    ACLD4     RA100               1.   RCLD4               1.
    BCLD4     R0000            .4978   RC0ST             .512
    BCLD4     2RB007             -1.   RCLD4               1.
    ECLD4     R0000           4.8185   RC0ST           2.6417
    ECLD4     RLDEX              -1.   RCLD4               1.
    ECLD4     RE039              -1.   RE104               1.
    HCLD4     R0000            .5773   RC0ST             .742
    HCLD4     RCLD4               1.   2RH039             -1.
    1CLD4     R0000            .3436   RC0ST            .6584
    1CLD4     RCLD4               1.   R1054              -1.
    JCLD4     R0000           1.6987   RC0ST            2.217
    JCLD4     RLDEX              -1.   RCLD4               1.
    JCLD4     1RJ008             -1.   1RJ050              1.
    ADLD4     R0000           4.3985   RC0ST           2.1042
    ADLD4     RLDEX              -1.   RA023              -1.
    ADLD4     RA100               1.   RDLD4               1.
    BDLD4     R0000            .2692   RC0ST            .4396
    BDLD4     2RB007             -1.   RDLD4               1.
    EDLD4     R0000           4.5113   RC0ST           2.3458
    EDLD4     RLDEX              -1.   RDLD4               1.
    EDLD4     RE039              -1.   RE104               1.
    HDLD4     R0000            .1697   RC0ST            .3354
    HDLD4     RDLD4               1.   2RH039             -1.
    1DLD4     R0000            .2737   RC0ST            .3358
    1DLD4     RDLD4               1.   R1054              -1.
    JDLD4     R0000           1.2169   RC0ST            1.825
    JDLD4     RLDEX              -1.   RDLD4               1.
    JDLD4     1RJ008             -1.   1RJ050              1.
    BJW37     R0000            .3251   RC0ST             .579
    BJW37     RB007             -.45   RB008             -.55
    BJW37     RJ036               1.   UKLDF              -1.
    1BJW37    R0000            .3251   RC0ST             .579
    1BJW37    RB007             -.45   1RB008            -.55
    1BJW37    RJ036               1.   UKLDF              -1.
    YJW37     R0000           1.5376   RC0ST            2.579
    YJW37     RLDEX              -1.   RJ036               1.
    YJW37     RY001               1.
    BNATG     R0000            .5656   RC0ST             .808
    BNATG     RLDEX               1.   RB039              -1.
    BNATG     RN009               1.
    BXATG     R0000            .4622   RC0ST             .725
    BXATG     RLDEX               1.   RB039              -1.
    BXATG     RXATG               1.
    HBK19     R0000            .3197   RC0ST             .629
    HBK19     RB041               1.   RH041              -1.
    JBK19     R0000            1.301   RC0ST            2.038
    JBK19     RPKEX              -1.   RB041               1.
    JBK19     RJ039              -1.
    BDK19     R0000            .2691   RC0ST              .44
    BDK19     RB041              -1.   RDK19               1.
    HDK19     R0000            .1698   RC0ST             .335
    HDK19     RDK19               1.   RH041              -1.
    1DK19     R0000            .2737   RC0ST             .336
    1DK19     RDK19               1.   R1058              -1.
    JDK19     R0000           1.2169   RC0ST            1.825
    JDK19     RPKEX              -1.   RDK19               1.
    JDK19     RJ039              -1.
    BHK19     R0000            .2875   RC0ST               .5
    BHK19     RB041              -1.   RH041               1.
    BHK19     RH106               1.
    1HK19     R0000             .256   RC0ST             .377
    1HK19     RH041               1.   RH106               1.
    1HK19     R1058              -1.
    JHK19     R0000           1.2071   RC0ST            1.883
    JHK19     RPKEX              -1.   RH041               1.
    JHK19     RH106               1.   RJ039              -1.
    H1K19     R0000            .3102   RC0ST             .492
    H1K19     RH041              -1.   R1058               1.
    J1K19     R0000           1.4161   RC0ST            2.112
    J1K19     RPKEX              -1.   R1058               1.
    J1K19     RJ039              -1.
    BNK19     R0000            .5233   30PGK               1.
    BNK19     RC0ST             .808   RPKEX               1.
    BNK19     RB041              -1.
    HNK19     R0000            .3624   30PGK               1.
    HNK19     RC0ST             .575   RPKEX               1.
    HNK19     RH041              -1.
    1NK19     R0000            .5357   30PGK               1.
    1NK19     RC0ST             .729   RPKEX               1.
    1NK19     R1058              -1.
    JNK19     R0000           1.5557   30PGK               1.
    JNK19     RC0ST            2.429   RH106               1.
    JNK19     RJ039              -1.
    JAATK     R0000          -4.5305   RC0ST            1.688
    JAATK     RA041               1.   RA100              -1.
    JAATK     RJ038              -1.
    ABATK     R0000           6.1787   RC0ST            2.133
    ABATK     RRKEX              -1.   RA041              -1.
    ABATK     RA100               1.   RB042               1.
    JBATK     R0000            1.301   RC0ST            2.038
    JBATK     RRKEX              -1.   RB042               1.
    JBATK     RJ038              -1.
    ZBATK     R0000           1.7531   RC0ST            2.708
    ZBATK     RRKEX              -1.   RB042               1.
    ZBATK     RZ012               1.
    AHATK     R0000           6.1875   RC0ST           2.1958
    AHATK     RRKEX              -1.   RA041              -1.
    AHATK     RA100               1.   2RH042              1.
    AHATK     RH106               1.
    BHATK     R0000            .2875   RC0ST               .5
    BHATK     RB042              -1.   RB098               1.
    BHATK     2RH042              1.   RH106               1.
    1HATK     R0000             .256   RC0ST             .377
    1HATK     2RH042              1.   R1057              -1.
    JHATK     R0000           1.2071   RC0ST            1.883
    JHATK     RRKEX              -1.   2RH042              1.
    JHATK     RH106               1.   RJ038              -1.
    ZHATK     R0000            1.659   RC0ST            2.554
    ZHATK     RRKEX              -1.   2RH042              1.
    ZHATK     RH106               1.   RZ012               1.
    A1ATK     R0000           6.3686   RC0ST            2.367
    A1ATK     RRKEX              -1.   RA041              -1.
    A1ATK     RA100               1.   R1057               1.
    B1ATK     R0000            .4671   RC0ST             .717
    B1ATK     RB042              -1.   RB098               1.
    B1ATK     R1057               1.
    J1ATK     R0000           1.4161   RC0ST            2.112
    J1ATK     RRKEX              -1.   R1057               1.
    J1ATK     RJ038              -1.
    BJATK     R0000             .301   RC0ST             .292
    BJATK     RRKEX               1.   RB042              -1.
    BJATK     RB098               1.   RJ084               1.
    ZJATK     R0000           1.5604   RC0ST           1.9333
    ZJATK     RJ084               1.   RZ012               1.
    BNATK     R0000            .5656   30ATK               1.
    BNATK     RC0ST             .808   RRKEX               1.
    BNATK     RB042              -1.   RB098               1.
    HNATK     R0000            .3624   30ATK               1.
    HNATK     RC0ST             .575   RRKEX               1.
    HNATK     2RH042             -1.
    1NATK     R0000            .5357   30ATK               1.
    1NATK     RC0ST             .729   RRKEX               1.
    1NATK     R1057              -1.
    JNATK     R0000            1.598   30ATK               1.
    JNATK     RC0ST            2.429   RH106               1.
    JNATK     RJ038              -1.
    ANG17     R0000           5.2133   30G17               1.
    ANG17     RC0ST            2.725   RA043              -1.
    ANG17     RA075              .57   RA076             -.59
    ANG17     RA077             -3.5   RA079               6.
    ANG17     RA081             -15.   RA100               1.
    ANG17     RH106               1.
    BNG17     R0000            .4919   30G17               1.
    BNG17     RC0ST             .808   RG0EX               1.
    BNG17     RB045              -1.   RB078              -1.
    HNG17     R0000            .3435   30G17               1.
    HNG17     RC0ST             .575   RG0EX               1.
    HNG17     RH043              -1.   RH080            1.506
    HNG17     RH081           -4.146
    1NG17     R0000            .5168   30G17               1.
    1NG17     RC0ST             .729   RG0EX               1.
    1NG17     R1060              -1.   R1088            -4.08
    JNG17     R0000           1.5368   30G17               1.
    JNG17     RC0ST            2.429   RH106               1.
    JNG17     RJ040              -1.   RJ061              -.8
    JNG17     RJ064               6.
    MNG17     R0000             .858   30G17               1.
    MNG17     RC0ST            1.646   RH106               1.
    MNG17     1RM003             -1.
    NNG17     R0000           -.3435   30G17              -1.
    NNG17     RC0ST            -.575   RG0EX              -1.
    NNG17     1RN014              1.
    HAG20     R0000          -3.0715   RC0ST              1.8
    HAG20     RG0EX               1.   RA044               1.
    HAG20     RA100              -1.   RH046           -1.002
    HAG20     RH062             .002   1RH089          -1.434
    HAG20     2RH089              4.   3RH089           8.434
    AEG20     R0000           1.3172   KW1RC              -1.
    AEG20     RA044              -1.   RE044               1.
    1AJG22    R0000           3.8065   RC0ST               .5
    1AJG22    RA043              -1.   RA077             -3.5
    1AJG22    RA081             -15.   RA100               1.
    1AJG22    RJ040               1.
    BJG22     R0000            .3011   RC0ST            .2917
    BJG22     RG0EX               1.   RB047              -1.
    BJG22     RB082             -1.8   RB083              2.5
    BJG22     RB086            -7.42   1RJ084              1.
    MJG22     R0000            .2747   RC0ST             .329
    MJG22     1RJ084              1.   1RM003             -1.
    YJG22     R0000            .4078   RC0ST             .858
    YJG22     RJ040               1.   RJ079               1.
    YJG22     RY002               1.
    1YJG22    R0000           2.0671   RC0ST            3.338
    1YJG22    1RJ084              1.   RY002               1.
    ZJG22     R0000            .5866   RC0ST             .954
    ZJG22     RJ040               1.   RJ079               1.
    ZJG22     RZ016               1.
    ANG22     R0000           5.2133   30G22               1.
    ANG22     RC0ST            2.725   RA043              -1.
    ANG22     RA077             -3.5   RA081             -15.
    ANG22     RA100               1.   RH106               1.
    BNG22     R0000            .4919   30G22               1.
    BNG22     RC0ST             .808   RG0EX               1.
    BNG22     RB047              -1.   RB082             -1.8
    BNG22     RB086            -7.42
    HNG22     R0000            .3435   30G22               1.
    HNG22     RC0ST             .575   RG0EX               1.
    HNG22     RH048              -1.
    1NG22     R0000            .5168   30G22               1.
    1NG22     RC0ST             .729   RG0EX               1.
    1NG22     R1062              -1.   R1093            -9.48
    JNG22     R0000           1.5368   30G22               1.
    JNG22     RC0ST            2.429   RH106               1.
    JNG22     RJ040              -1.
    1MNG22    R0000             .858   30G22               1.
    1MNG22    RC0ST           1.6458   RH106               1.
    1MNG22    2RM003             -1.
    YNG22     R0000           1.8099   30G22               1.
    YNG22     RC0ST            3.004   RH106               1.
    YNG22     RY002               1.
    H1G27     R0000            .3102   RC0ST             .492
    H1G27     RH046              -1.   R1061               1.
    1ARG29    R0000           5.0009   RC0ST             2.45
    1ARG29    RA043              -1.   RA075            -1.42
    1ARG29    RA077              -1.   RA078             -.99
    1ARG29    RA079              1.5   RA081             -15.
    1ARG29    RA100               1.   RH106              .33
    1ARG29    RRG29               1.
    1BRG29    R0000            .2829   RC0ST             .517
    1BRG29    RG0EX               1.   RB047              -1.
    1BRG29    RB081              -.6   RB082              7.2
    1BRG29    RB083               4.   RB084               .5
    1BRG29    RB085            -2.19   RB086            -3.76
    1BRG29    RRG29               1.
    1HRG29    R0000             .192   RC0ST             .404
    1HRG29    RG0EX               1.   RH043              -1.
    1HRG29    RH080            2.245   RH081          -12.606
    1HRG29    RH082               5.   RH083              -5.
    1HRG29    RH084           -1.379   RRG29               1.
    11RG29    R0000            .3039   RC0ST             .429
    11RG29    RG0EX               1.   R1062              -1.
    11RG29    R1091             -.73   R1092               4.
    11RG29    R1094            -1.98   R1095              -5.
    11RG29    R1096              25.   RRG29               1.
    1JRG29    R0000            1.217   RC0ST            1.904
    1JRG29    RH106              .25   RJ011             -.13
    1JRG29    RJ014              -.2   RJ015               .2
    1JRG29    RJ018             -.98   RJ019              .11
    1JRG29    RJ031            -.022   RJ079               1.
    1JRG29    RJ100               .5   RRG29               1.
    2JRG29    R0000            1.217   RC0ST            1.904
    2JRG29    RH106              .25   RJ011             -.13
    2JRG29    RJ014              -.2   RJ015             -.31
    2JRG29    RJ016              .04   RJ018             -.47
    2JRG29    RJ019              .07   RJ031            -.037
    2JRG29    RJ079               1.   RJ100              .67
    2JRG29    RRG29               1.
    1MRG29    R0000            .3003   RC0ST            1.104
    1MRG29    2RM003             -1.   RRG29               1.
    ANG30     R0000           5.2133   30G30               1.
    ANG30     RC0ST            2.725   RA043              -1.
    ANG30     RA075              .57   RA076             -.59
    ANG30     RA079               6.   RA081             -15.
    ANG30     RA100               1.   RH106               1.
    BNG30     R0000            .4919   30G30               1.
    BNG30     RC0ST             .808   RG0EX               1.
    BNG30     RB045              -1.   RB077               5.
    BNG30     RB078              -1.   RB100               1.
    HNG30     R0000            .3435   30G30               1.
    HNG30     RC0ST             .575   RG0EX               1.
    HNG30     RH043              -1.   RH080            1.506
    HNG30     RH081           -4.146   RH082               5.
    HNG30     RH083              -3.
    1NG30     R0000            .5168   30G30               1.
    1NG30     RC0ST             .729   RG0EX               1.
    1NG30     R1060              -1.   R1087               5.
    1NG30     R1088            -4.08
    BRG33     R0000            .2987   RC0ST              .55
    BRG33     RG0EX               1.   RB047              -1.
    BRG33     RB081              -.6   RB082              7.2
    BRG33     RB083              2.5   RB084               .5
    BRG33     RB085            -2.19   RB086            -5.23
    BRG33     RR010               1.
    HRG33     R0000            .2098   RC0ST             .442
    HRG33     RG0EX               1.   RH043              -1.
    HRG33     RH080            2.245   RH081          -10.909
    HRG33     RH082               3.   RH083              -5.
    HRG33     RH084           -1.379   RR010               1.
    1RG33     R0000            .3216   RC0ST             .467
    1RG33     RG0EX               1.   R1062              -1.
    1RG33     R1091            -1.47   R1092              2.5
    1RG33     R1093             3.15   R1094            -2.65
    1RG33     R1095               5.   R1096              25.
    1RG33     RR010               1.
    JRG33     R0000            1.217   RC0ST            1.904
    JRG33     RJ040              -1.   RJ062              2.5
    JRG33     RJ063            -2.53   RJ064               3.
    JRG33     RJ065               5.   RR010               1.
    2MRG33    R0000            .3162   RC0ST           1.1375
    2MRG33    2RM003             -1.   RR010               1.
    1HNG44    R0000            .3435   30G44               1.
    1HNG44    RC0ST             .575   RG0EX               1.
    1HNG44    2RH043             -1.
    11NG44    R0000            .5168   30G44               1.
    11NG44    RC0ST             .729   RG0EX               1.
    11NG44    R1060              -1.   R1086             -1.5
    11NG44    R1087               5.   R1088              .61
    BDG55     R0000            .2566   RC0ST              .44
    BDG55     4RB045             -1.   RDG55               1.
    HDG55     R0000            .1698   RC0ST             .335
    HDG55     RDG55               1.   3RH048             -1.
    1DG55     R0000            .2737   RC0ST             .336
    1DG55     RDG55               1.   2R1062             -1.
    JDG55     R0000           1.2169   RC0ST            1.825
    JDG55     RG0EX              -1.   RDG55               1.
    JDG55     1RJ041             -1.
    MDG55     R0000            .5461   RC0ST            1.058
    MDG55     RG0EX              -1.   RDG55               1.
    MDG55     2RM003             -1.
    BHG55     R0000             .275   RC0ST               .5
    BHG55     4RB045             -1.   3RH048              1.
    BHG55     RH106               1.
    1HG55     R0000             .256   RC0ST             .377
    1HG55     3RH048              1.   RH106               1.
    1HG55     2R1062             -1.
    JHG55     R0000           1.2071   RC0ST            1.883
    JHG55     RG0EX              -1.   3RH048              1.
    JHG55     RH106               1.   1RJ041             -1.
    MHG55     R0000            .5362   RC0ST            1.117
    MHG55     RG0EX              -1.   3RH048              1.
    MHG55     RH106               1.   2RM003             -1.
    B1G55     R0000            .4546   RC0ST             .717
    B1G55     4RB045             -1.   2R1062              1.
    H1G55     R0000            .3102   RC0ST             .492
    H1G55     3RH048             -1.   2R1062              1.
    J1G55     R0000           1.4161   RC0ST            2.112
    J1G55     RG0EX              -1.   2R1062              1.
    J1G55     1RJ041             -1.
    M1G55     R0000             .745   RC0ST            1.346
    M1G55     RG0EX              -1.   2R1062              1.
    M1G55     2RM003             -1.
    YAD11     R0000          -2.5103   RC0ST            2.025
    YAD11     RA046               1.   RY002              .89
    YAD11     RY004              .11
    ZAD11     R0000          -2.3364   RC0ST            2.121
    ZAD11     RA046               1.   RZ016              .65
    ZAD11     RZ019              .35
    AED16     R0000           1.3106   KW1RC              -1.
    AED16     1RA047             -1.   RE047               1.
    MHD20     R0000            .4804   RC0ST            1.117
    MHD20     RF0EX              -1.   2RH025              1.
    MHD20     RH106               1.   RM004              -1.
    AJD20     R0000           3.8249   RC0ST            .5917
    AJD20     RA046              -1.   RA082            -4.06
    AJD20     RA083             1.39   RA086              -9.
    AJD20     RA087              12.   RA088              -6.
    AJD20     RJ043               1.
    JVD20     R0000            .9756   RC0ST           1.5042
    JVD20     RJ043              -1.   RV001               1.
    YVD20     R0000           1.2182   RC0ST            2.075
    YVD20     RV001               1.   RY002               1.
    ZVD20     R0000           1.2774   RC0ST           2.1333
    ZVD20     RV001               1.   RZ016               1.
    BND22     R0000            .4621   30D22               1.
    BND22     RC0ST             .833   RG0EX               1.
    BND22     3RB047             -1.
    HND22     R0000            .3254   30D22               1.
    HND22     RC0ST               .6   RG0EX               1.
    HND22     2RH048             -1.
    1ND22     R0000             .491   30D22               1.
    1ND22     RC0ST             .754   RG0EX               1.
    1ND22     2R1063             -1.
    EJD26     R0000           3.8023   RC0ST               .6
    EJD26     RE046              -1.   RJ043               1.
    BND30     R0000            .4621   30D30               1.
    BND30     RC0ST             .833   RG0EX               1.
    BND30     2RB047             -1.
    HND30     R0000            .3254   30D30               1.
    HND30     RC0ST               .6   RG0EX               1.
    HND30     2RH048             -1.
    1ND30     R0000             .491   30D30               1.
    1ND30     RC0ST             .754   RG0EX               1.
    1ND30     1R1063             -1.
    ATF07     R0000           2.9319   RA051              -1.
    ATF07     RA093            -3.19   RA094              3.1
    ATF07     RA095              -6.   RA096              -3.
    ATF07     RT005               1.
    ETF07     R0000           3.0015   RE048              -1.
    ETF07     RE088             1.04   RE089             1.14
    ETF07     RE091              18.   RT005               1.
    JTF07     R0000           1.0134   RJ045              -1.
    JTF07     RJ073             1.05   1RJ073            -.94
    JTF07     1RJ074             11.   RT005               1.
    YTF07     R0000           1.1479   RT005               1.
    YTF07     RY002             .046   RY004             .954
    ATF09     R0000           2.9319   RA048              -1.
    ATF09     RA089             -1.2   RA090              .81
    ATF09     RA091             -2.5   RA092               4.
    ATF09     RT006               1.
    ETF09     R0000           3.0015   1RE056             -1.
    ETF09     RT006               1.
    JTF09     R0000           1.0134   RJ042              -1.
    JTF09     RJ102              -3.   RJ103             -.15
    JTF09     RJ104             -.69   RJ105             -.31
    JTF09     RJ106               6.   RT006               1.
    ZTF09     R0000           1.1756   RT006               1.
    ZTF09     RZ020               1.
    HNF21     R0000            .3148   30121               .5
    HNF21     30F52               .5   RC0ST             .575
    HNF21     RF0EX               1.   RH051              -1.
    HNF21     RH091              13.
    1NF21     R0000            .4804   30121               .9
    1NF21     RC0ST             .729   RF0EX               1.
    1NF21     R1018             -.32   R1033             -.15
    1NF21     1R1033            -.53   R1065               .1
    1NF21     R1106             -.75   R1107              .65
    1NF21     R1108             -.29   R1109             -.11
    1NF21     R1110             -4.9
    HBF28     R0000            .2033   RC0ST             .429
    HBF28     RB050               1.   1RH051             -1.
    BHF28     R0000            .2292   RC0ST               .5
    BHF28     RB050              -1.   1RH051              1.
    BHF28     RH106               1.
    B1F28     R0000             .398   RC0ST             .717
    B1F28     RB050              -1.   1R1064              1.
    H1F28     R0000            .3335   RC0ST             .604
    H1F28     1RH051             -1.   1R1064              1.
    HBF35     R0000            .2033   RC0ST             .429
    HBF35     RB052               1.   3RH052             -1.
    JBF35     R0000           1.1142   RC0ST            1.838
    JBF35     RF0EX              -1.   RB052               1.
    JBF35     2RJ046             -1.
    MBF35     R0000            .4716   RC0ST            1.071
    MBF35     RF0EX              -1.   RB052               1.
    MBF35     RM005              -1.
    BCF35     R0000            .3721   RC0ST             .512
    BCF35     RB052              -1.   RCF35               1.
    HCF35     R0000            .4234   RC0ST             .742
    HCF35     RCF35               1.   3RH052             -1.
    JCF35     R0000           1.4671   RC0ST            2.217
    JCF35     RF0EX              -1.   RCF35               1.
    JCF35     2RJ046             -1.
    MCF35     R0000            .8382   RC0ST            1.458
    MCF35     RF0EX              -1.   RCF35               1.
    MCF35     RM005              -1.
    ZCF35     R0000           1.8837   RC0ST            2.883
    ZCF35     RF0EX              -1.   RCF35               1.
    ZCF35     RZ019               1.
    BDF35     R0000            .4905   RC0ST            .6792
    BDF35     RB052              -1.   RDF35               1.
    HDF35     R0000             .153   RC0ST             .335
    HDF35     RDF35               1.   3RH052             -1.
    1DF35     R0000            .3125   RC0ST             .467
    1DF35     RDF35               1.   R1069              -1.
    JDF35     R0000           1.1257   RC0ST            1.825
    JDF35     RF0EX              -1.   RDF35               1.
    JDF35     2RJ046             -1.
    MDF35     R0000            .4932   RC0ST            1.058
    MDF35     RF0EX              -1.   RDF35               1.
    MDF35     RM005              -1.
    ZDF35     R0000           1.3948   RC0ST           2.4583
    ZDF35     RF0EX              -1.   RDF35               1.
    ZDF35     RZ019               1.
    BHF35     R0000            .2292   RC0ST               .5
    BHF35     RB052              -1.   3RH052              1.
    BHF35     RH106               1.
    1HF35     R0000             .293   RC0ST             .509
    1HF35     3RH052              1.   RH106               1.
    1HF35     R1069              -1.
    JHF35     R0000           1.1129   RC0ST            1.883
    JHF35     RF0EX              -1.   3RH052              1.
    JHF35     RH106               1.   2RJ046             -1.
    MHF35     R0000            .4804   RC0ST            1.117
    MHF35     RF0EX              -1.   3RH052              1.
    MHF35     RH106               1.   RM005              -1.
    ZHF35     R0000           1.3821   RC0ST           2.5167
    ZHF35     RF0EX              -1.   3RH052              1.
    ZHF35     RH106               1.   RZ019               1.
    B1F35     R0000             .398   RC0ST             .717
    B1F35     RB052              -1.   R1069               1.
    H1F35     R0000            .3335   RC0ST             .604
    H1F35     3RH052             -1.   R1069               1.
    J1F35     R0000           1.3104   RC0ST            2.112
    J1F35     RF0EX              -1.   R1069               1.
    J1F35     2RJ046             -1.
    M1F35     R0000            .6778   RC0ST            1.346
    M1F35     RF0EX              -1.   R1069               1.
    M1F35     RM005              -1.
    Z1F35     R0000           1.5795   RC0ST           2.7458
    Z1F35     RF0EX              -1.   R1069               1.
    Z1F35     RZ019               1.
    MJF35     R0000            .2582   RC0ST            .3292
    MJF35     RJ046               1.   RM005              -1.
    YJF35     R0000           1.1795   RC0ST           1.8958
    YJF35     RJ046               1.   RY002              .13
    YJF35     RY004              .87
    HVF35     R0000            .2531   RC0ST             .458
    HVF35     RF0EX               1.   3RH052             -1.
    HVF35     RV002               1.
    JVF35     R0000           1.0101   RC0ST            1.504
    JVF35     2RJ046             -1.   RV002               1.
    YVF35     R0000           1.2527   RC0ST            2.075
    YVF35     RV002               1.   RY002              .13
    YVF35     RY004              .87
    ZVF35     R0000           1.2774   RC0ST           2.1333
    ZVF35     RV002               1.   RZ019               1.
    YJF46     R0000            .3649   RC0ST             .858
    YJF46     RJ045               1.   RJ072               1.
    YJF46     RJ073             -.28   1RJ073            -.53
    YJF46     1RJ074            -14.   RJ080               1.
    YJF46     RY004               1.
    JVF46     R0000            .9756   RC0ST            1.504
    JVF46     RJ045              -1.   RV003               1.
    BNF48     R0000            .4908   30F48               1.
    BNF48     RC0ST              .95   RF0EX               1.
    BNF48     1RB053             -1.   RB102             3.38
    BNF48     RB103            -3.11   RB104              12.
    BNF48     RB106             -7.5
    HNF48     R0000            .3148   30F48               1.
    HNF48     RC0ST             .575   RF0EX               1.
    HNF48     3RH053             -1.
    1NF48     R0000            .4804   30F48               1.
    1NF48     RC0ST             .729   RF0EX               1.
    1NF48     R1069              -1.   R1111             1.06
    1NF48     R1112            -1.67   R1115               2.
    JNF48     R0000           1.4154   30F48               1.
    JNF48     RC0ST            2.429   RH106               1.
    JNF48     2RJ046             -1.   RJ076             1.03
    JNF48     1RJ076           -1.77   RJ078               2.
    ZNF48     R0000           1.6187   30F48               1.
    ZNF48     RC0ST            2.675   RZ020               1.
    BNF52     R0000            .4908   30F52               1.
    BNF52     RC0ST              .95   RF0EX               1.
    BNF52     RB050              -1.
    HNF52     R0000            .3148   30F52               1.
    HNF52     RC0ST             .575   RF0EX               1.
    HNF52     1RH051             -1.
    1NF52     R0000            .4804   30F52               1.
    1NF52     RC0ST             .729   RF0EX               1.
    1NF52     1R1064             -1.
    ZJF57     R0000            .3896   RC0ST            .9167
    ZJF57     RJ089               1.   RZ019               1.
    EAF59     R0000           1.1876   KW1RC               1.
    EAF59     RA049               1.   RE058              -1.
    EAF59     RE098            -1.82   RE099             2.76
    EAF59     RE100               .5   RE101              25.
    EAF60     R0000           1.1876   KW1RC               1.
    EAF60     RA051               1.   RE048              -1.
    EAF60     RE088             4.12   RE089            -2.85
    EAF60     RE090               7.   RE091              21.
    AEF60     R0000           1.3106   KW1RC              -1.
    AEF60     RA051              -1.   RE050               1.
    EAF64     R0000           1.1876   KW1RC               1.
    EAF64     RA052               1.   RE048              -1.
    EAF64     RE088             1.27   RE090               6.
    EAF64     RE091              10.
    B1F64     R0000             .398   RC0ST             .717
    B1F64     RB051            -.704   1RB053           -.296
    B1F64     RB091              .06   RB092              .29
    B1F64     1RB093            4.22   RB102               1.
    B1F64     RB103             -.92   RB104             2.96
    B1F64     RB106            -2.66   1R1068              1.
    H1F64     R0000            .3335   RC0ST             .604
    H1F64     2RH052             -1.   RH093             1.25
    H1F64     RH110            -1.06   1R1068              1.
    J1F64     R0000           1.3104   RC0ST            2.112
    J1F64     RF0EX              -1.   1R1068              1.
    J1F64     2RJ045             -1.   RJ073             1.27
    J1F64     1RJ073            -.78
    BNF64     R0000            .4908   2SF145              1.
    BNF64     RC0ST              .95   RF0EX               1.
    BNF64     RB051            -.704   1RB053           -.296
    BNF64     RB091              .06   RB092              .29
    BNF64     1RB093            4.22   RB102               1.
    BNF64     RB103             -.92   RB104             2.96
    BNF64     RB106            -2.66
    HNF64     R0000            .3148   2SF145              1.
    HNF64     RC0ST             .575   RF0EX               1.
    HNF64     2RH052             -1.   RH093             1.25
    HNF64     RH110            -1.06
    1NF64     R0000            .4804   2SF145              1.
    1NF64     RC0ST             .729   RF0EX               1.
    1NF64     1R1068             -1.
    JNF64     R0000           1.4154   2SF145              1.
    JNF64     RC0ST            2.429   RH106               1.
    JNF64     RJ045              -1.   RJ073             1.27
    JNF64     1RJ073            -.78
    YNF64     R0000           1.6597   2SF145              1.
    YNF64     RC0ST            3.004   RH106               1.
    YNF64     RY002             .055   RY004             .945
    AEF68     R0000           1.3106   KW1RC              -1.
    AEF68     RA013            -.236   RA019            -.764
    AEF68     RE049               1.
    AXF72     R0000           3.0796   RC0ST            1.308
    AXF72     1RA053             -1.   RXF72               1.
    EXF72     R0000           3.0963   RC0ST             .812
    EXF72     RE051              -1.   RXF72               1.
    1BF75     R0000            .3366   RC0ST             .562
    1BF75     RB053               1.   R1065              -1.
    1BF75     R1106              .46   R1107              .25
    HBF80     R0000            .2033   RC0ST             .429
    HBF80     2RB051              1.   2RH052             -1.
    JBF80     R0000           1.1142   RC0ST            1.838
    JBF80     RF0EX              -1.   2RB051              1.
    JBF80     2RJ045             -1.
    MBF80     R0000            .4716   RC0ST            1.071
    MBF80     RF0EX              -1.   2RB051              1.
    MBF80     RM006              -1.
    BCF80     R0000            .3721   RC0ST             .512
    BCF80     2RB051             -1.   RCF80               1.
    HCF80     R0000            .4234   RC0ST             .742
    HCF80     RCF80               1.   2RH052             -1.
    JCF80     R0000           1.4671   RC0ST            2.217
    JCF80     RF0EX              -1.   RCF80               1.
    JCF80     2RJ045             -1.
    MCF80     R0000            .8382   RC0ST            1.458
    MCF80     RF0EX              -1.   RCF80               1.
    MCF80     RM006              -1.
    BHF80     R0000            .2292   RC0ST               .5
    BHF80     2RB051             -1.   2RH052              1.
    BHF80     RH106               1.
    JHF80     R0000           1.1129   RC0ST            1.883
    JHF80     RF0EX              -1.   2RH052              1.
    JHF80     RH106               1.   2RJ045             -1.
    MHF80     R0000            .4804   RC0ST            1.117
    MHF80     RF0EX              -1.   2RH052              1.
    MHF80     RH106               1.   RM006              -1.
    BNF80     R0000            .4908   2SF129              1.
    BNF80     RC0ST              .95   RF0EX               1.
    BNF80     1RB051             -1.
    YJF83     R0000            .4078   RC0ST             .858
    YJF83     RJ080               1.   RJ091               1.
    YJF83     RY002             .025   RY004             .975
    HNF89     R0000            .3126   2SF089              1.
    HNF89     RC0ST             .575   RF0EX               1.
    HNF89     RH052              -1.   RH093             1.25
    HNF89     1RH093            -16.   RH094              21.
    HNF89     RH110            -1.06
    1NF89     R0000            .4804   2SF089              1.
    1NF89     RC0ST             .729   RF0EX               1.
    1NF89     R1065              -1.   R1106              .52
    1NF89     R1107             -.35   R1108              -5.
    117FHR    R0000            .1718   RC0ST            .4042
    117FHR    RH053              -1.   RH115              .24
    117FHR    RH117              -5.   RH118              15.
    117FHR    RR021               1.
    117FZR    R0000           1.3821   RC0ST           2.5167
    117FZR    RR021               1.   RZ019               1.
    118FAT    R0000           2.9319   RA048              -1.
    118FAT    RA091               .5   RA092               4.
    118FAT    3RT006              1.
    118FET    R0000           3.0015   RE048              -1.
    118FET    RE088             4.32   RE089            -1.96
    118FET    RE090              17.   RE091              16.
    118FET    3RT006              1.
    118FJT    R0000           1.0134   RJ042              -1.
    118FJT    RJ103             1.23   RJ104            -1.77
    118FJT    RJ106               6.   3RT006              1.
    118FZT    R0000           1.1756   3RT006              1.
    118FZT    RZ020               1.
    129FHB    R0000            .2033   RC0ST             .429
    129FHB    1RB051              1.   2RH051             -1.
    129FJB    R0000           1.1142   RC0ST            1.838
    129FJB    RF0EX              -1.   1RB051              1.
    129FJB    2RJ045             -1.
    129FBH    R0000            .2292   RC0ST               .5
    129FBH    1RB051             -1.   2RH051              1.
    129FBH    RH106               1.
    129F1H    R0000             .293   RC0ST             .509
    129F1H    2RH051              1.   RH106               1.
    129F1H    R1068              -1.
    129FJH    R0000           1.1129   RC0ST            1.883
    129FJH    RF0EX              -1.   2RH051              1.
    129FJH    RH106               1.   2RJ045             -1.
    HN129     R0000            .3148   2SF129              1.
    HN129     RC0ST             .575   RF0EX               1.
    HN129     RH052              -1.
    JN129     R0000           1.4154   2SF129              1.
    JN129     RC0ST            2.429   RH106               1.
    JN129     RJ045              -1.   RJ072               5.
    MN129     R0000            .7757   2SF129              1.
    MN129     RC0ST            1.646   RH106               1.
    MN129     RM006              -1.
    YN129     R0000           1.6597   2SF129              1.
    YN129     RC0ST           3.0042   RH106               1.
    YN129     RY004               1.
    HV129     R0000            .2531   RC0ST             .458
    HV129     RF0EX               1.   RH052              -1.
    HV129     RV003               1.
    YV129     R0000           1.2527   RC0ST            2.075
    YV129     RV003               1.   RY004               1.
    AT137     R0000           2.9319   RA051              -1.
    AT137     RA093              1.7   RA094            -2.87
    AT137     RA095               5.   RA096               9.
    AT137     4RT006              1.
    ET137     R0000           3.0015   RE058              -1.
    ET137     RE099              .78   RE100             -10.
    ET137     RE101              19.   4RT006              1.
    JT137     R0000           1.0134   RJ014              -.3
    JT137     RJ025              -.7   4RT006              1.
    YAFLD     R0000          -2.2771   RC0ST            2.025
    YAFLD     RA009              .87   RA010              .13
    YAFLD     RA060               1.   3RA060              1.
    YAFLD     RA100              -1.   RY001            1.007
    1YAFLD    R0000          -2.4796   RA009              .87
    1YAFLD    RA010              .13   RA055            2.133
    1YAFLD    2RA060              1.   3RA060              1.
    1YAFLD    RY001             1.01
    YEFLD     R0000           -1.987   RC0ST           2.5667
    YEFLD     3RA060              1.   RE012             .257
    YEFLD     RE013             .556   RE015             .187
    YEFLD     RE060             -.01   RE062               1.
    YEFLD     RE104              -1.   RY001             1.01
    1YEFLD    R0000           -2.238   3RA060              1.
    1YEFLD    RE005               1.   RE012             .257
    1YEFLD    RE013             .556   RE015             .187
    1YEFLD    RE060           -.0308   RE061              2.8
    1YEFLD    RE062               1.   RY001             1.01
    YHFLD     R0000            .9886   RH010             .285
    YHFLD     RH016             .665   RH096             .028
    YHFLD     RH105               1.   RH106               1.
    YHFLD     RY001             1.01
    1YHFLD    R0000           1.6603   RC0ST            2.458
    1YHFLD    RLDEX              -1.   RH007              .04
    1YHFLD    RH008              .08   1RH038             .66
    1YHFLD    1RH054              1.   RH057             1.15
    1YHFLD    RH096             -.12   RH106               1.
    1YHFLD    RH113              .12   RH114              .12
    1YHFLD    RY001               1.
    YJFLD     R0000            .4078   RC0ST             .858
    YJFLD     1RJ008              .3   RJ010              .55
    YJFLD     RJ012              .15   RJ099              .81
    YJFLD     RY001             1.01
    JASRB     R0000          -2.3645   RC0ST            1.688
    JASRB     RA009              1.6   RA023              -.6
    JASRB     RA060               1.   3RA060              1.
    JASRB     RA100              -1.   RJ010              -1.
    1JASRB    R0000          -2.3645   RC0ST            1.688
    1JASRB    RA009              1.6   RA023              -.6
    1JASRB    RA060               1.   3RA060              1.
    1JASRB    RA100              -1.   RJ009              -1.
    JESRB     R0000          -2.0691   RC0ST            2.238
    JESRB     3RA060              1.   RE013               1.
    JESRB     RE060             -.01   RE062               1.
    JESRB     RE104              -1.   RJ010              -1.
    1JESRB    R0000          -2.0691   RC0ST            2.238
    1JESRB    3RA060              1.   RE014               1.
    1JESRB    RE060             -.01   RE062               1.
    1JESRB    RE104              -1.   RJ009              -1.
    BHSRB     R0000            .2875   RC0ST               .5
    BHSRB     RB008              -1.   RH015              .95
    BHSRB     RH096             .025   RH105            .9494
    BHSRB     RH106               1.
    1BHSRB    R0000            .2875   RC0ST               .5
    1BHSRB    1RB008             -1.   RH016              .95
    1BHSRB    RH096             .025   RH105            .9494
    1BHSRB    RH106               1.
    2BHSRB    R0000            .2875   RC0ST               .5
    2BHSRB    RB008              -1.   RH015              .99
    2BHSRB    1RH054             .99   RH106               1.
    3BHSRB    R0000            .2875   RC0ST               .5
    3BHSRB    1RB008             -1.   RH016              .99
    3BHSRB    1RH054             .99   RH106               1.
    JHSRB     R0000           1.1129   RC0ST            1.883
    JHSRB     RLDEX              -1.   RH016              .95
    JHSRB     RH096             .025   RH105            .9494
    JHSRB     RH106               1.   RJ010              -1.
    1JHSRB    R0000           1.1129   RC0ST            1.883
    1JHSRB    RLDEX              -1.   RH015              .95
    1JHSRB    RH096             .025   RH105            .9494
    1JHSRB    RH106               1.   RJ009              -1.
    BHFCR     R0000            .2875   RC0ST               .5
    BHFCR     1RB024             -1.   1RH038              1.
    JHFCR     R0000           1.2071   RC0ST            1.883
    JHFCR     RMSEX              -1.   1RH038              1.
    JHFCR     RH106               1.   RJ097              -1.
    1JHFCR    R0000           1.3921   RC0ST            1.883
    1JHFCR    RMSEX              -1.   RH035               1.
    1JHFCR    RH067            .1307   RH068            7.843
    1JHFCR    RH069          -17.017   RH070            7.937
    1JHFCR    RH071           -.3259   RH106               1.
    1JHFCR    RJ098              -1.
    2JHFCR    R0000           1.3921   RC0ST            1.883
    2JHFCR    RMSEX              -1.   RH035               1.
    2JHFCR    RH067            2.604   RH068             .643
    2JHFCR    RH069          -10.936   RH070             6.43
    2JHFCR    RH071            -.317   RH106               1.
    2JHFCR    RJ097              -.4   1RJ097             -.6
    3JHFCR    R0000           1.3921   RC0ST            1.883
    3JHFCR    RMSEX              -1.   RH035               1.
    3JHFCR    RH067            3.267   RH068            7.843
    3JHFCR    RH069          -17.017   RH070            7.937
    3JHFCR    RH071           -.3259   RH106               1.
    3JHFCR    RJ098              -.4   1RJ098             -.6
    ZHFCR     R0000           1.8441   RC0ST            2.554
    ZHFCR     RMSEX              -1.   RH035               1.
    ZHFCR     RH067            -.128   RH068            1.282
    ZHFCR     RH069          -13.305   RH070             6.41
    ZHFCR     RH071            -.316   RH106               1.
    ZHFCR     RZ004               1.
    ZJFCR     R0000            .7736   RC0ST             .954
    ZJFCR     RJ032               1.   RJ052             -4.5
    ZJFCR     RJ053             10.3   RJ054            -10.2
    ZJFCR     RJ095            -.402   RZ004               1.
    H1HCR     R0000            .3102   RC0ST             .492
    H1HCR     RH011            -.548   RH095            -.452
    H1HCR     R1043              -1.
    BHLCS     R0000            .2875   RC0ST               .5
    BHLCS     RB028              -1.   RH012               1.
    BHLCS     RH106               1.
    1HHSC     R0000             .425   RC0ST             .377
    1HHSC     RH035               1.   RH067            -.708
    1HHSC     RH068        69.649899   RH069           -19.95
    1HHSC     RH070        82.639899   RH071            -.291
    1HHSC     RH106               1.   R1041               1.
    11HHSC    R0000             .334   RC0ST             .377
    11HHSC    RH036               1.   RH072           -14.99
    11HHSC    RH073           -19.95   RH074              -5.
    11HHSC    RH075            -.529   RH106               1.
    11HHSC    R1041               1.
    1HLSC     R0000             .454   RC0ST             .377
    1HLSC     RH035               1.   RH067            1.379
    1HLSC     RH068            -63.4   RH069            27.31
    1HLSC     RH070           -34.48   RH071             -.34
    1HLSC     RH106               1.   R1040               1.
    11HLSC    R0000             .347   RC0ST             .377
    11HLSC    RH036               1.   RH072           -16.14
    11HLSC    RH073            27.31   RH074              -5.
    11HLSC    RH075            -.618   RH106               1.
    11HLSC    R1040               1.
    EAKER     R0000           1.1876   KW1RC               1.
    EAKER     RA013               1.   RE017              -1.
    EHKER     R0000           4.8104   RC0ST            2.221
    EHKER     RG0EX              -1.   RE017              -1.
    EHKER     RE104               1.   RH048               1.
    EHKER     RH085            3.944   RH086          -28.786
    EHKER     RH087             -6.8   RH088             -14.
    EHKER     RH089            2.987   RH106               1.
    EAALG     R0000           1.1876   KW1RC               1.
    EAALG     RA014               1.   RE019              -1.
    EHALG     R0000           4.8104   RC0ST            2.221
    EHALG     RG0EX              -1.   RE019              -1.
    EHALG     RE104               1.   RH048               1.
    EHALG     RH085           -2.306   RH086          -10.107
    EHALG     RH087             -1.5   RH088             -4.5
    EHALG     RH089           -2.729   RH106               1.
    EAKLG     R0000           1.1876   KW1RC               1.
    EAKLG     RA015               1.   RE018              -1.
    EHKLG     R0000           4.8104   RC0ST            2.221
    EHKLG     RG0EX              -1.   RE018              -1.
    EHKLG     RE104               1.   RH048               1.
    EHKLG     RH085           -1.452   RH086           -10.18
    EHKLG     RH087              1.3   RH088              -9.
    EHKLG     RH089           -2.795   RH106               1.
    MUKG0     R0000           -.9481   RC0ST           -1.342
    MUKG0     1RM003             -1.   RY002              -1.
    1YBLG0    R0000           1.5743   RC0ST            2.612
    1YBLG0    RG0EX              -1.   RB047               1.
    1YBLG0    RB081            3.565   RB082             -5.4
    1YBLG0    RB083               .3   RB084              -.9
    1YBLG0    RB085            -.636   RB086             2.16
    1YBLG0    RY002               1.
    2YBLG0    R0000           1.5743   RC0ST            2.612
    2YBLG0    RG0EX              -1.   RB046               1.
    2YBLG0    1RB050              1.   RB087            -14.9
    2YBLG0    RB088            -22.2   RB089            -.636
    2YBLG0    RB090             -25.   1RB090            -18.
    2YBLG0    RY002               1.
    3YBLG0    R0000           1.5743   RC0ST            2.612
    3YBLG0    RG0EX              -1.   RB051               1.
    3YBLG0    RB091           -22.57   RB092            13.08
    3YBLG0    1RB092            -54.   RB093            -.636
    3YBLG0    1RB093           -30.2   RY002               1.
    1ZBLG0    R0000           1.7531   RC0ST            2.708
    1ZBLG0    RG0EX              -1.   RB047               1.
    1ZBLG0    RB081            1.053   RB082             -3.8
    1ZBLG0    RB083             -1.4   RB084              -.5
    1ZBLG0    RB085            -.636   RB086             2.16
    1ZBLG0    RZ016               1.
    2ZBLG0    R0000           1.7531   RC0ST            2.708
    2ZBLG0    RG0EX              -1.   RB046               1.
    2ZBLG0    1RB050              1.   RB087            -14.9
    2ZBLG0    RB088            -23.9   RB089            -.636
    2ZBLG0    RB090             -25.   1RB090            -18.
    2ZBLG0    RZ016               1.
    3ZBLG0    R0000           1.7531   RC0ST            2.708
    3ZBLG0    RG0EX              -1.   RB051               1.
    3ZBLG0    RB091           -22.57   RB092            13.08
    3ZBLG0    1RB092            -54.   RB093            -.636
    3ZBLG0    1RB093           -31.9   RZ016               1.
    YHLG0     R0000           1.4803   RC0ST            2.458
    YHLG0     RG0EX              -1.   RH051               1.
    YHLG0     RH090           -16.33   1RH090            -22.
    YHLG0     2RH090             15.   RH091            -22.2
    YHLG0     RH103               1.   RH106               1.
    YHLG0     RH108             -30.   RY002               1.
    1YHLG0    R0000           1.4803   RC0ST            2.458
    1YHLG0    RG0EX              -1.   RH052               1.
    1YHLG0    RH093           -24.18   1RH093            -61.
    1YHLG0    RH094            -30.2   RH106               1.
    1YHLG0    RH110            23.51   RY002               1.
    4YHLG0    R0000           1.4803   RC0ST            2.458
    4YHLG0    RG0EX              -1.   RH048               1.
    4YHLG0    RH086          -13.333   RH087              -.2
    4YHLG0    RH088             -10.   RH089              -.7
    4YHLG0    RH106               1.   RY002               1.
    YHLGG     R0000           1.4803   RC0ST            2.458
    YHLGG     RG0EX              -1.   RH106               1.
    YHLGG     RH109               1.   1RH109          -18.23
    YHLGG     2RH109           16.95   RY002               1.
    ZHLG0     R0000            1.659   RC0ST            2.554
    ZHLG0     RG0EX              -1.   RH051               1.
    ZHLG0     RH090           -16.33   1RH090            -17.
    ZHLG0     2RH090             15.   RH091            -23.9
    ZHLG0     RH103               1.   RH106               1.
    ZHLG0     RH108             -30.   RZ016               1.
    1ZHLG0    R0000            1.659   RC0ST            2.554
    1ZHLG0    RG0EX              -1.   RH052               1.
    1ZHLG0    RH093           -24.18   1RH093            -56.
    1ZHLG0    RH094            -31.9   RH106               1.
    1ZHLG0    RH110            23.51   RZ016               1.
    3ZHLG0    R0000            1.659   RC0ST            2.554
    3ZHLG0    RG0EX              -1.   RH048               1.
    3ZHLG0    RH085           -2.447   RH086            -6.77
    3ZHLG0    RH087             -1.9   RH088              -6.
    3ZHLG0    RH089              -.7   RH106               1.
    3ZHLG0    RZ016               1.
    EAAHG     R0000           1.1876   KW1RC               1.
    EAAHG     RA016               1.   RE022              -1.
    EAKHG     R0000           1.1876   KW1RC               1.
    EAKHG     RA017               1.   RE021              -1.
    EAHWD     R0000           1.1876   KW1RC               1.
    EAHWD     RA026               1.   RE028              -1.
    AJCG0     R0000           4.0551   RC0ST            1.425
    AJCG0     RA033              -1.   RA100               1.
    AJCG0     RJ040               1.   RJ061            -10.7
    AJCG0     RJ062               7.   RJ063            -3.43
    AJCG0     RJ064               4.   RJ065              29.
    1AJCG0    R0000           4.0551   RC0ST            1.425
    1AJCG0    RA033              -1.   RJ043               1.
    1AJCG0    RJ066             -7.8   RJ067               2.
    1AJCG0    RJ068            -3.43   RJ069              -6.
    1AJCG0    RJ070              -.5   RJ071              16.
    MHRES     R0000            .4804   RC0ST            1.117
    MHRES     RF0EX              -1.   RH052               1.
    MHRES     RH094              -5.   RH106               1.
    MHRES     RH110             -.67   RM006              -1.
    1MHRES    R0000            .4804   RC0ST            1.117
    1MHRES    RF0EX              -1.   RH052               1.
    1MHRES    RH093            -3.08   RH094              -5.
    1MHRES    RH106               1.   RH110             2.41
    1MHRES    RM005              -1.
    2MHRES    R0000            .4804   RC0ST            1.117
    2MHRES    RF0EX              -1.   RH052               1.
    2MHRES    RH106               1.   RH110              .67
    2MHRES    1RM006             -1.
    3MHRES    R0000            .4804   RC0ST            1.117
    3MHRES    RF0EX              -1.   RH052               1.
    3MHRES    RH106               1.   RH110              .67
    3MHRES    2RM006             -1.
    YHRES     R0000           1.3573   RC0ST            2.458
    YHRES     RF0EX              -1.   RH052               1.
    YHRES     1RH093             -7.   RH094              -.5
    YHRES     RH106               1.   RH110             -.67
    YHRES     RY004               1.
    ZHRES     R0000           1.5313   RC0ST            2.554
    ZHRES     RF0EX              -1.   RH052               1.
    ZHRES     RH093            -4.05   1RH093             -5.
    ZHRES     RH094             -15.   RH106               1.
    ZHRES     RH110             3.38   RZ019               1.
    ZJRES     R0000            .5388   RC0ST             .954
    ZJRES     RJ045               1.   RJ072             -16.
    ZJRES     RJ073            -4.46   1RJ073            3.65
    ZJRES     RJ074            -1.03   RJ080               1.
    ZJRES     RZ019               1.
    1ZJRES    R0000            .5388   RC0ST             .954
    1ZJRES    RJ046               1.   RJ075             -16.
    1ZJRES    RJ076            -1.34   1RJ076              .7
    1ZJRES    RJ077            -1.03   RJ078               3.
    1ZJRES    RJ080               1.   RZ019               1.
    2ZJRES    R0000            .5388   RC0ST             .954
    2ZJRES    RJ042               1.   RJ080               1.
    2ZJRES    RJ102              -1.   RJ103            -1.34
    2ZJRES    RJ104               .7   RJ105            -1.03
    2ZJRES    RJ106               3.   RZ019               1.
    ZJREX     R0000            .5388   RC0ST             .954
    ZJREX     RJ045               1.   RJ072             -16.
    ZJREX     RJ073            -4.46   1RJ073            3.65
    ZJREX     RJ074            -1.03   1RJ074             -7.
    ZJREX     RJ080               1.   RZ020               1.
    1ZJREX    R0000            .5388   RC0ST             .954
    1ZJREX    RJ046               1.   RJ075             -16.
    1ZJREX    RJ076            -1.34   1RJ076              .7
    1ZJREX    RJ077            -1.03   RJ078              -4.
    1ZJREX    RJ080               1.   RZ020               1.
    2ZJREX    R0000            .5388   RC0ST             .954
    2ZJREX    RJ042               1.   RJ080               1.
    2ZJREX    RJ102              -1.   RJ103            -1.34
    2ZJREX    RJ104               .7   RJ105            -1.03
    2ZJREX    RJ106              -4.   RZ020               1.
    HSPMS     R0000              -9.   1RH035             -1.
    HSPMS     HSPMS               1.
    HSRMS     R0000              -8.   1RH036             -1.
    HSRMS     HSRMS               1.
    HSLDF     R0000              -7.   RH039              -1.
    HSLDF     HSLDF               1.
    HSATK     R0000              -8.   2RH042             -1.
    HSATK     HSATK               1.
    NSG17     R0000             -6.5   1RN014             -1.
    NSG17     NSG17               1.
    HSG58     R0000              -6.   2RH048             -1.
    HSG58     HSG58               1.
    HSF80     R0000              -2.   2RH052             -1.
    HSF80     HSF80               1.
    2CB002    RB017             .511   RB001             .707
    2CB002    1RB003            .303   1RB005            .003
    2CB002    RB006             .009   RB007              .03
    2CB002    RB008             .102   1RB008            .023
    2CB002    RB009             .045   RB011             .129
    2CB002    RB013             .063   2RB015            .041
    2CB002    RB016             .017   RB022             .025
    2CB002    RB031            -.026   1RB031            .002
    2CB002    RB058               1.   1RB058              1.
    2CB002    RB101             .983
    3CB003    2RB020            .539   RB003             .707
    3CB003    1RB003            .303   1RB005            .002
    3CB003    RB006             .011   RB007             .057
    3CB003    RB008             .062   1RB008            .023
    3CB003    RB010             .071   RB012             .082
    3CB003    1RB013             .07   2RB015            .041
    3CB003    RB016             .016   RB022             .024
    3CB003    RB031            -.024   1RB031            .002
    3CB003    RB058               1.   1RB058              1.
    3CB003    RB101             1.02
    3CB004    2RB018            .484   1RB001            .707
    3CB004    1RB003            .303   RB006             .008
    3CB004    1RB007             .05   1RB008            .122
    3CB004    RB009             .047   RB011             .135
    3CB004    1RB013            .063   2RB015            .041
    3CB004    1RB016             .02   RB022             .029
    3CB004    RB031            -.025   1RB031            .001
    3CB004    RB058               1.   1RB058              1.
    3CB004    RB101             .983
    4CB004    2RB018            .484   1RB001            .707
    4CB004    1RB003            .303   RB006             .008
    4CB004    2RB007            .096   2RB008            .076
    4CB004    RB009             .047   RB011             .135
    4CB004    1RB013            .063   2RB015            .041
    4CB004    1RB016             .02   RB022             .029
    4CB004    RB031            -.025   1RB031            .001
    4CB004    RB058               1.   1RB058              1.
    4CB004    RB099             .096   RB101             .983
    4CB004    UKLDF            -.096
    3CB005    1RB019            .553   RB002             .707
    3CB005    1RB003            .303   1RB005            .003
    3CB005    RB006              .01   1RB007            .045
    3CB005    1RB008            .094   RB009             .039
    3CB005    RB011             .108   1RB013            .058
    3CB005    2RB015            .041   1RB016             .02
    3CB005    RB022             .026   RB031            -.024
    3CB005    1RB031            .003   RB058               1.
    3CB005    1RB058              1.   RB101            1.005
    1CB096    2RB018             -1.   RB053               1.
    1CB096    RB094             -1.5   RB095              .79
    1CB096    RB096              4.7   RB097              3.2
    2CB096    1RB019             -1.   RB053               1.
    2CB096    RB094             -.02   RB095             -.69
    2CB096    RB096              8.4   RB097             -4.6
    3CB096    2RB020             -1.   RB053               1.
    3CB096    RB094              .39   RB095             -1.1
    3CB096    RB096             -1.6   RB097              4.4
    2CB095    1RB019             -1.   1RB053              1.
    2CB095    RB102             -.02   RB103            -1.73
    2CB095    RB104              1.4   RB105            -.803
    2CB095    RB106              3.4
    3CB095    2RB020             -1.   1RB053              1.
    3CB095    RB102              .39   RB103            -2.14
    3CB095    RB104             10.4   RB105            -.803
    3CB095    RB106             -6.6
    CB096     RB017              -1.   RB053               1.
    CB096     RB094            -1.55   RB095              .84
    CB096     RB096             -4.3   RB097              5.5
    3CB084    2RB020             -1.   RB051               1.
    3CB084    RB091             -.37   RB092              -.8
    3CB084    1RB092            20.4   RB093            -.803
    3CB084    1RB093           -21.6
    CB095     RB017              -1.   1RB053              1.
    CB095     RB102            -1.55   RB103              -.2
    CB095     RB104             11.5   RB105            -.803
    CB095     RB106             -9.3
    1CB095    2RB018             -1.   1RB053              1.
    1CB095    RB102             -1.5   RB103             -.25
    1CB095    RB104              9.2   RB105            -.803
    1CB095    RB106              -.3
    CB084     RB017              -1.   RB051               1.
    CB084     RB091            -2.31   RB092             1.14
    CB084     1RB092            21.5   RB093            -.803
    CB084     1RB093           -24.3
    1CB084    2RB018             -1.   RB051               1.
    1CB084    RB091            -2.26   RB092             1.09
    1CB084    1RB092            19.2   RB093            -.803
    1CB084    1RB093           -15.3
    2CB084    1RB019             -1.   RB051               1.
    2CB084    RB091             -.78   RB092             1.09
    2CB084    1RB092            11.4   RB093            -.803
    2CB084    1RB093           -11.6
    4CB005    1RB019            .553   RB002             .707
    4CB005    1RB003            .303   1RB005            .003
    4CB005    RB006              .01   2RB007             .08
    4CB005    2RB008            .059   RB009             .039
    4CB005    RB011             .108   1RB013            .058
    4CB005    2RB015            .041   1RB016             .02
    4CB005    RB022             .026   RB031            -.024
    4CB005    1RB031            .003   RB058               1.
    4CB005    1RB058              1.   RB099              .08
    4CB005    RB101            1.005   UKLDF             -.08
    1CB068    RB009             -.07   RB011             -.24
    1CB068    RB012             -.32   RB015              -.3
    1CB068    1RB016            -.07   1RB045              1.
    CH458     3RH030             -1.   RH109               1.
    CH458     1RH109            9.43   2RH109          -10.91
    CH463     2RH030             -1.   RH109               1.
    CH463     1RH109           12.61   2RH109          -14.09
    CB152     RB022              -1.   RB031           1.0313
    CB153     1RB022             -1.   RB031           1.0148
    1C1076    R1019             -.86   R1022             -.14
    1C1076    R1062               1.   R1091            -2.28
    1C1076    R1092              2.6   R1093              9.3
    1C1076    R1094            -3.17   R1095              -5.
    1C1078    R1020             -.88   R1023             -.12
    1C1078    R1062               1.   R1091            -1.68
    1C1078    R1092             -1.7   R1093             9.38
    1C1078    R1094            -3.17   R1095              -6.
    C1080     R1021             -.92   R1024             -.08
    C1080     R1062               1.   R1091            -5.88
    C1080     R1092             -7.2   R1093             9.05
    C1080     R1094            -3.24   R1095               7.
    CH147     2RH025             -1.   RH096               1.
    3CH151    1RH029             -1.   RH096               1.
    CH526     RH060               1.   1RH060             -1.
    1CJ015    R0000             .201   RJ010              -1.
    1CJ015    RJ032               1.   1RJ050              1.
    1CJ015    RJ052              42.   RJ053            30.29
    1CJ015    RJ054           -42.14
    2CJ015    R0000             .201   RJ009              -1.
    2CJ015    RJ032               1.   1RJ050              1.
    2CJ015    RJ052              42.   RJ053            30.29
    2CJ015    RJ054           -42.14
    1CJ062    RJ018              -1.   RJ031            -.055
    1CJ062    RJ040               1.   RJ061             -3.8
    1CJ062    RJ062              -3.   RJ063            -3.01
    1CJ062    RJ064               7.   RJ065             -6.5
    1CJ062    RJ079               1.   RJ100               1.
    128FNN    30F48              -1.   30128               1.
    90AS30    R0000            .2079   30G22              -1.
    90AS30    RC0ST            .4375   RH048               1.
    90A930    R0000             .186   2SF129             -1.
    90A930    RC0ST            .4375   RH052               1.
    3CB007    1RB003            .303   1RB004            .707
    3CB007    RB006             .006   1RB007            .035
    3CB007    1RB008            .064   RB010             .059
    3CB007    RB012             .069   1RB015            .218
    3CB007    2RB015             .02   RB022             .038
    3CB007    2RB025             .49   RB031            -.026
    3CB007    1RB031            .001   RB058               1.
    3CB007    1RB058              1.   RB101            1.015
    4CB007    1RB003            .303   1RB004            .707
    4CB007    RB006             .006   2RB007            .069
    4CB007    2RB008             .03   RB010             .059
    4CB007    RB012             .069   RB015             .218
    4CB007    2RB015             .02   RB022             .038
    4CB007    2RB025             .49   RB031            -.026
    4CB007    1RB031            .001   RB058               1.
    4CB007    1RB058              1.   RB099             .069
    4CB007    RB101            1.015   UKLDF            -.069
    1CH301    RH001            1.455   1RH006            1.01
    1CH301    RH007             .003   RH008            .0103
    1CH301    1RH008            .002   RH010            .0444
    1CH301    RH015             .097   RH018            .0475
    1CH301    RH019            .1101   2RH023           .0707
    1CH301    2RH025            .005   1RH027            .505
    1CH301    1RH029            .102   RH096            -.017
    1CH301    RH107            1.455   1RH107             -1.
    1CH301    2RH107              1.   RH113             .003
    2CH301    RH001            1.455   1RH006            1.01
    2CH301    RH007             .003   RH008            .0103
    2CH301    1RH008            .002   RH010            .0444
    2CH301    RH015             .097   RH018            .0475
    2CH301    RH019            .1101   2RH023           .0707
    2CH301    2RH025            .005   1RH027            .217
    2CH301    2RH028           .0058   1RH029           .2518
    2CH301    4RH030           .1324   RH060             .288
    2CH301    RH096            -.027   RH107            1.455
    2CH301    1RH107             -1.   2RH107              1.
    2CH301    RH113             .003
    3CH301    2RH025             -1.   RH029               1.
    1CH119    RH026              -1.   1RH028             .07
    1CH119    RH029             .459   RH030             .471
    1CH119    1RH060              1.   RH096            -.027
    3CH119    2RH017             -1.   1RH028             .08
    3CH119    RH029             .537   3RH030            .383
    3CH119    1RH060              1.   RH096            -.027
    1CH121    RH027              -1.   1RH028            .113
    1CH121    RH029             .525   RH030             .362
    1CH121    1RH060           1.007   RH096            -.027
    C1129     R1030              -1.   R1039             1.03
    1C1129    R1031              -1.   R1039             1.04
    1ZHREX    R0000           1.5313   RC0ST            2.554
    1ZHREX    RH053               1.   RH115             -.48
    1ZHREX    RH116              -1.   RH117               1.
    1ZHREX    RH118             -14.   1RH118             -5.
    1ZHREX    RZ020               1.
RHS
    .00001    2SF145             29.   2SF089             60.
    .00001    2SF129             73.   30M00              77.
    .00001    30M94              27.   30M91              44.
    .00001    30ATK               4.   30PGK              23.
    .00001    30G30              44.   30G17             164.
    .00001    30G44              31.   30G22              79.
    .00001    30D22              19.   30D30               1.
    .00001    30121               4.   30F52              86.
    .00001    30F48              88.   30128              20.
    .00001    RCRFT            -62.5   RA009               .8
    .00001    RA023               2.   RA035             2.42
    .00001    RA036            12.13   RA037             12.2
    .00001    RA038              4.6   RA040               1.
    .00001    RA041              31.   RA043              25.
    .00001    1RA043              7.   RA044              24.
    .00001    RA046               8.   RA047               5.
    .00001    RA049               2.   RA051        68.299998
    .00001    RA052              32.   1RA052              5.
    .00001    RA054              19.   RA055             700.
    .00001    1RA055            350.   RA056             135.
    .00001    RA057             230.   RA058             115.
    .00001    1RA058             45.   RA059              60.
    .00001    RA060              35.   2RA060             25.
    .00001    3RA060            133.   RA097              80.
    .00001    1RA099             28.   RB002             215.
    .00001    2RB005             3.6   1RB006             6.4
    .00001    2RB007             22.   1RB022             12.
    .00001    2RB030              5.   RB031               4.
    .00001    1RB031              5.   RB033               9.
    .00001    RB035              60.   RB037              31.
    .00001    RB038              10.   1RB038             52.
    .00001    RB039               6.   RB040              6.5
    .00001    1RB040             1.2   2RB040             1.1
    .00001    RB041               7.   RB042              22.
    .00001    1RB043              .8   1RB044             1.8
    .00001    1RB045             51.   4RB045             33.
    .00001    RB046              60.   3RB047             30.
    .00001    RB048             24.7   2RB049              4.
    .00001    RB050              57.   2RB051            113.
    .00001    RB052              23.   RB053              89.
    .00001    RB054              28.   1RB054             28.
    .00001    RB055              14.   RB057             96.7
    .00001    RB058      1321.899979   1RB058      881.299988
    .00001    RB059       182.099997   RB060       110.599999
    .00001    RB061       130.299997   RB062             23.5
    .00001    RB063              22.   RB098              18.
    .00001    RB099              50.   RB101      1376.199966
    .00001    RCLD4              34.   RCF35              14.
    .00001    RCF80              98.   RDBMS              16.
    .00001    RDLD4              11.   RDK19              55.
    .00001    RDG55               1.   RDF35              83.
    .00001    RE005              67.   RE008              4.3
    .00001    RE009              4.8   RE010              1.5
    .00001    RE011              1.8   RE036              56.
    .00001    RE040              17.   RE044               7.
    .00001    RE046              11.   RE047              18.
    .00001    RE049               5.   RE050              72.
    .00001    RE055              23.   RE057              17.
    .00001    RE060             -4.7   RE061             400.
    .00001    1RE061           -160.   2RE061            280.
    .00001    RE062             120.   RE063             105.
    .00001    RE064              35.   RE066             9.34
    .00001    RE103               4.   RH001      1972.699966
    .00001    RH008             19.4   1RH008             3.2
    .00001    RH009              -1.   RH010               1.
    .00001    RH011               4.   RH016            -14.2
    .00001    RH018               .2   2RH025            -16.
    .00001    RH031             44.3   RH032              14.
    .00001    1RH034              8.   1RH035             42.
    .00001    1RH036             22.   RH038             2.68
    .00001    2RH038            4.03   2RH039             30.
    .00001    RH041              14.   RH042              30.
    .00001    2RH042             48.   RH046              73.
    .00001    3RH048             38.   RH050              30.
    .00001    1RH051             89.   2RH051             12.
    .00001    2RH052      177.699997   3RH052             49.
    .00001    4RH052             50.   2RH053             16.
    .00001    3RH053            26.3   RH054             4.88
    .00001    1RH054             80.   RH055             3.66
    .00001    RH056              60.   RH057       323.299995
    .00001    RH058              98.   RH059            18.34
    .00001    RH060       390.399994   RH061       134.199999
    .00001    RH062             1.02   RH096             3.97
    .00001    RH099       109.799999   RH103             200.
    .00001    RH105            45.75   RH106            2000.
    .00001    1RH106             60.   2RH106            300.
    .00001    RH107       986.399986   2RH107            260.
    .00001    RH109              73.   RH114             18.8
    .00001    R1003             175.   2R1006            130.
    .00001    R1009             4.71   1R1009             1.2
    .00001    R1010              4.1   R1012              8.7
    .00001    R1039             54.9   1R1039             35.
    .00001    R1040             23.2   R1041              11.
    .00001    R1042             -10.   R1043              -.6
    .00001    R1045               6.   R1047              52.
    .00001    R1051             27.8   R1055              62.
    .00001    R1057              15.   1R1057             15.
    .00001    R1058               3.   R1061              36.
    .00001    2R1062             49.   2R1063              8.
    .00001    1R1064             24.   R1065              61.
    .00001    R1066              10.   1R1068            154.
    .00001    R1069              44.   R1070              14.
    .00001    R1071       635.899986   1R1071            793.
    .00001    2R1071           823.5   R1072       231.799999
    .00001    R1073            129.9   R1074             16.5
    .00001    RJ001             200.   RJ003              24.
    .00001    RJ007               .9   1RJ007              4.
    .00001    RJ008               5.   RJ031              6.6
    .00001    RJ032              3.6   RJ033              6.4
    .00001    RJ034             160.   RJ036              44.
    .00001    1RJ036             42.   1RJ037             17.
    .00001    2RJ037             -2.   RJ038              31.
    .00001    RJ039               2.   RJ040              22.
    .00001    RJ043              66.   RJ045              56.
    .00001    RJ046             193.   RJ047              17.
    .00001    RJ048      1432.299973   RJ049              69.
    .00001    RJ050             78.9   1RJ050       97.599999
    .00001    RJ051               9.   RJ079             320.
    .00001    RJ080             180.   RJ081               1.
    .00001    RJ084              24.   1RJ084              8.
    .00001    RJ089              10.   RJ091             193.
    .00001    RJ092               4.   RJ100             13.5
    .00001    1RM001            208.   RM011              4.5
    .00001    RM012              4.5   RN009              18.
    .00001    RR010              30.   RR021              68.
    .00001    RRG29              81.   HSPMS             100.
    .00001    HSRMS             100.   HSLDF              50.
    .00001    HSATK              30.   NSG17             100.
    .00001    HSG58             100.   HSF80             100.
    .00001    RT005              85.   RT006              15.
    .00001    3RT006              9.   4RT006             15.
    .00001    RV001              10.   RV002              22.
    .00001    RV003               4.   RXATG              16.
    .00001    RXF72              26.   RY001               7.
    .00001    1RY001             29.   RY005             518.
    .00001    RZ002               4.   RZ004              19.
    .00001    RZ005              23.   RZ012              43.
    .00001    RZ016              63.   RZ019              30.
    .00001    RZ020             118.   RABDH             506.
    .00001    RAGJA             575.   RGACH             966.
    .00001    RHN1G             436.   RK1RK             272.
    .00001    RL1BY             200.   RMN1G              79.
    .00001    RMURB             148.   RQATR             146.
    .00001    RT1JA             247.
ENDATA
